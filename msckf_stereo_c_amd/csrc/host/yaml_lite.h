// yaml_lite.h — the subset of yaml-cpp's interface the reference uses (YAML::LoadFile, Node::operator[],
// Node::as<T>), enough to read config/camchain-imucam-euroc.yaml, app_imgproc.yaml and
// app_msckfvio.yaml (keys listed in SURVEY.md Appendix D).  yaml-cpp itself is not installed here.
// Supported: block mappings nested by indentation, scalars, flow sequences of scalars ("[a, b, c]",
// possibly spanning several lines), '#' comments.  Conversions mirror
// msckf_core/include/common/config_io.h:13-81 (Vector3/Vector4 from 3/4-lists, Mat4 from a row-major 16-list).
#pragma once
#include <cstdlib>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace YAML {

class Exception : public std::runtime_error {
  public:
    explicit Exception(const std::string &m) : std::runtime_error(m) {}
};

class Node {
  public:
    Node() : d_(std::make_shared<Data>()) {}
    bool IsDefined() const { return d_->defined; }
    Node operator[](const std::string &key) const {
        auto it = d_->map.find(key);
        if (it == d_->map.end()) { Node n; n.d_->path = d_->path + "/" + key; return n; }
        return it->second;
    }
    template <typename T> T as() const;
    const std::vector<std::string> &seq() const { need(); return d_->seq; }
    const std::string &scalar() const { need(); return d_->scalar; }

    // builder interface (parser only)
    void set_scalar(const std::string &s) { d_->defined = true; d_->scalar = s; }
    void set_seq(const std::vector<std::string> &v) { d_->defined = true; d_->seq = v; d_->is_seq = true; }
    Node &child(const std::string &k) { d_->defined = true; Node &n = d_->map[k]; n.d_->path = d_->path + "/" + k; return n; }

  private:
    struct Data {
        bool defined = false, is_seq = false;
        std::string scalar, path;
        std::vector<std::string> seq;
        std::map<std::string, Node> map;
    };
    void need() const { if (!d_->defined) throw Exception("yaml: missing key " + d_->path); }
    std::shared_ptr<Data> d_;
};

namespace detail {
inline std::string trim(const std::string &s) {
    size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}
inline std::vector<std::string> split_seq(const std::string &body) {
    std::vector<std::string> out;
    std::stringstream ss(body);
    std::string item;
    while (std::getline(ss, item, ',')) { item = trim(item); if (!item.empty()) out.push_back(item); }
    return out;
}
}  // namespace detail

inline Node Load(std::istream &in) {
    // comment-stripped, non-empty lines
    std::vector<std::string> lines;
    {
        std::string line;
        while (std::getline(in, line)) {
            size_t hash = line.find('#');
            if (hash != std::string::npos) line = line.substr(0, hash);
            if (!detail::trim(line).empty()) lines.push_back(line);
        }
    }
    Node root;
    root.child("__root__");  // mark defined
    struct Level { int indent; Node node; };
    std::vector<Level> stack;
    stack.push_back({-1, root});
    for (size_t li = 0; li < lines.size(); ++li) {
        const std::string &line = lines[li];
        int indent = (int)line.find_first_not_of(' ');
        std::string t = detail::trim(line);
        size_t colon = t.find(':');
        if (colon == std::string::npos) throw Exception("yaml: expected 'key: value' in '" + t + "'");
        std::string key = detail::trim(t.substr(0, colon));
        std::string val = detail::trim(t.substr(colon + 1));
        // a flow sequence may start on the following line ("key:" newline "[ ... ]")
        if (val.empty() && li + 1 < lines.size() && detail::trim(lines[li + 1])[0] == '[') val = detail::trim(lines[++li]);
        // gather a multi-line flow sequence
        if (!val.empty() && val[0] == '[') {
            while (val.find(']') == std::string::npos) {
                if (li + 1 >= lines.size()) throw Exception("yaml: unterminated '[' for key " + key);
                val += " " + detail::trim(lines[++li]);
            }
        }
        while (stack.size() > 1 && stack.back().indent >= indent) stack.pop_back();
        Node &parent = stack.back().node;
        Node &n = parent.child(key);
        if (val.empty()) {
            n.child("__map__");
            stack.push_back({indent, n});
        } else if (val[0] == '[') {
            size_t e = val.find(']');
            n.set_seq(detail::split_seq(val.substr(1, e - 1)));
        } else {
            n.set_scalar(val);
        }
    }
    return root;
}

inline Node LoadFile(const std::string &path) {
    std::ifstream f(path);
    if (!f.good()) throw Exception("yaml: cannot open " + path);
    return Load(f);
}

template <> inline std::string Node::as<std::string>() const { return scalar(); }
template <> inline double Node::as<double>() const { return std::strtod(scalar().c_str(), nullptr); }
template <> inline int Node::as<int>() const { return (int)std::strtod(scalar().c_str(), nullptr); }
template <> inline std::vector<double> Node::as<std::vector<double>>() const {
    std::vector<double> v;
    for (const auto &s : seq()) v.push_back(std::strtod(s.c_str(), nullptr));
    return v;
}

}  // namespace YAML
