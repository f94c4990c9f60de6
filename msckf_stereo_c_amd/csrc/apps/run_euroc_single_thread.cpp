// run_euroc_single_thread — headless counterpart of the reference harness apps/run_euroc_single_thread.cpp.
//
// Same argument (<path-to-euroc-mav0-dir>), same config path ("../config/camchain-imucam-euroc.yaml", Q16),
// same CSV parsing (timestamp = stoi(seconds)*1e9 + stoi(nanoseconds) in double, IMU values via stof; Q9),
// same call order per image: do { imu_callback } while (t_imu <= t_img); stereo_callback; backend_callback
// (reference :189-254, Q10).  No OpenCV / Pangolin: images are decoded by a small zlib-based reader of
// 8-bit grayscale PNG (what EuRoC ships) or binary PGM, and nothing is drawn (Q17).  Writes pose_out.txt.
#include <zlib.h>
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>
#include "../host/system.h"

static bool read_file(const std::string &path, std::vector<unsigned char> &buf) {
    std::ifstream f(path, std::ios::binary);
    if (!f.good()) return false;
    buf.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    return true;
}

static unsigned be32(const unsigned char *p) { return (unsigned)p[0] << 24 | (unsigned)p[1] << 16 | (unsigned)p[2] << 8 | p[3]; }

// 8-bit grayscale, non-interlaced PNG -> YImg8
static bool decode_png(const std::vector<unsigned char> &d, cg::YImg8 &out) {
    static const unsigned char sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (d.size() < 33 || memcmp(d.data(), sig, 8) != 0) return false;
    size_t pos = 8;
    unsigned w = 0, h = 0;
    std::vector<unsigned char> idat;
    while (pos + 12 <= d.size()) {
        const unsigned len = be32(&d[pos]);
        const std::string type((const char *)&d[pos + 4], 4);
        const unsigned char *body = &d[pos + 8];
        if (pos + 12 + len > d.size()) return false;
        if (type == "IHDR") {
            w = be32(body); h = be32(body + 4);
            if (body[8] != 8 || body[9] != 0 || body[12] != 0) { std::cerr << "PNG: only 8-bit gray non-interlaced is supported\n"; return false; }
        } else if (type == "IDAT") idat.insert(idat.end(), body, body + len);
        else if (type == "IEND") break;
        pos += 12 + len;
    }
    if (!w || !h) return false;
    std::vector<unsigned char> raw((size_t)h * (w + 1));
    uLongf rawlen = raw.size();
    if (uncompress(raw.data(), &rawlen, idat.data(), idat.size()) != Z_OK || rawlen != raw.size()) return false;
    out = cg::YImg8((int)h, (int)w);
    unsigned char *img = out.data();
    for (unsigned y = 0; y < h; ++y) {
        const unsigned char ft = raw[(size_t)y * (w + 1)];
        const unsigned char *src = &raw[(size_t)y * (w + 1) + 1];
        unsigned char *row = img + (size_t)y * w;
        const unsigned char *up = y ? row - w : nullptr;
        for (unsigned x = 0; x < w; ++x) {
            const int a = x ? row[x - 1] : 0, b = up ? up[x] : 0, c = (x && up) ? up[x - 1] : 0;
            int pred = 0;
            switch (ft) {
                case 0: pred = 0; break;
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) >> 1; break;
                case 4: { const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
                default: return false;
            }
            row[x] = (unsigned char)(src[x] + pred);
        }
    }
    return true;
}

static bool decode_pgm(const std::vector<unsigned char> &d, cg::YImg8 &out) {
    if (d.size() < 15 || d[0] != 'P' || d[1] != '5') return false;
    std::string hdr((const char *)d.data(), std::min<size_t>(d.size(), 64));
    std::istringstream ss(hdr);
    std::string magic; int w, h, mx;
    ss >> magic >> w >> h >> mx;
    const size_t off = (size_t)ss.tellg() + 1;
    if (mx != 255 || off + (size_t)w * h > d.size()) return false;
    out = cg::YImg8(h, w);
    memcpy(out.data(), d.data() + off, (size_t)w * h);
    return true;
}

static bool load_gray(const std::string &path, cg::YImg8 &out) {
    std::vector<unsigned char> buf;
    if (!read_file(path, buf)) return false;
    return decode_png(buf, out) || decode_pgm(buf, out);
}

int main(int argc, char *argv[]) {
    if (argc != 2) {
        std::cout << "Arguments ERROR!" << std::endl;
        std::cout << "Usage: run_xxx <path-to-euroc-mav0-dir>" << std::endl;
        return -1;
    }
    const std::string euroc_dir = argv[1];
    const std::string file_cam_imu = "../config/camchain-imucam-euroc.yaml";
    const int num_cams = 2;
    cg::System system(file_cam_imu);
    if (!system.ok()) { std::cerr << "ERROR: cannot initialise the system (config files / GPU)" << std::endl; return -1; }

    std::vector<std::vector<std::pair<double, std::string>>> data_img(num_cams);
    for (int n = 0; n < num_cams; ++n) {
        std::ifstream cam_file(euroc_dir + "/cam" + std::to_string(n) + "/data.csv");
        if (!cam_file.good()) { std::cerr << "ERROR: no cam file found !!!" << std::endl; return -1; }
        std::string line;
        std::getline(cam_file, line);
        while (std::getline(cam_file, line)) {
            std::stringstream stream(line);
            std::string s;
            std::getline(stream, s, ',');
            const std::string nanoseconds = s.substr(s.size() - 9, 9), seconds = s.substr(0, s.size() - 9);
            const double stamp_ns = std::stoi(seconds) * 1e9 + std::stoi(nanoseconds);
            std::getline(stream, s, ',');
            std::string imgname = s;
            while (!imgname.empty() && (imgname.back() == '\r' || imgname.back() == '\n' || imgname.back() == ' ')) imgname.pop_back();
            data_img[n].push_back(std::make_pair(stamp_ns, imgname));
        }
    }
    assert(data_img[0].size() == data_img[1].size());
    std::ifstream imu_file(euroc_dir + "/imu0/data.csv");
    if (!imu_file.good()) { std::cerr << "ERROR: no imu file found !!!" << std::endl; return -1; }
    std::string line;
    std::getline(imu_file, line);

    const size_t imgs_size = data_img[0].size();
    for (size_t idx_img = 0; idx_img < imgs_size; ++idx_img) {
        cg::Image imgs[2];
        bool ok = true;
        for (int j = 0; j < num_cams; ++j) {
            imgs[j].time_stamp = data_img[j][idx_img].first * 1e-9;
            const std::string img_path = euroc_dir + "/cam" + std::to_string(j) + "/data/" + data_img[j][idx_img].second;
            if (!load_gray(img_path, imgs[j].image) || imgs[j].image.empty()) { std::cerr << "ERROR: img is empty !!! " << img_path << std::endl; ok = false; }
        }
        if (!ok) return -1;
        const double t_img = imgs[0].time_stamp;
        double t_imu = 0.0;
        do {
            if (!std::getline(imu_file, line)) { t_imu = 1e300; break; }
            std::stringstream stream(line);
            std::string s;
            std::getline(stream, s, ',');
            const std::string nanoseconds = s.substr(s.size() - 9, 9), seconds = s.substr(0, s.size() - 9);
            const double stamp_ns = std::stoi(seconds) * 1e9 + std::stoi(nanoseconds);
            cg::Vector3 gyr, acc;
            for (int j = 0; j < 3; ++j) { std::getline(stream, s, ','); gyr[j] = std::stof(s); }
            for (int j = 0; j < 3; ++j) { std::getline(stream, s, ','); acc[j] = std::stof(s); }
            std::shared_ptr<cg::Imu> imu(new cg::Imu);
            imu->time_stamp = stamp_ns * 1e-9;
            imu->angular_velocity = gyr;
            imu->linear_acceleration = acc;
            system.imu_callback(imu);
            t_imu = imu->time_stamp;
        } while (t_imu <= t_img);
        system.stereo_callback(imgs[0], imgs[1], false);
        system.backend_callback();
    }
    std::cout << "done: " << imgs_size << " stereo frames, " << system.path_to_draw_.size() << " poses in pose_out.txt" << std::endl;
    return 0;
}
