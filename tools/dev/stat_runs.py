"""Distribution of the headline over repeated bench runs: tools/dev/stat_runs.py gpurun_out/<prefix>_*.json (grouped by the prefix before the last _N)."""
import json, sys, collections, statistics
g = collections.defaultdict(list)
for f in sys.argv[1:]:
    try:
        d = json.load(open(f))
    except Exception:
        continue
    key = f.rsplit("_", 1)[0].split("/")[-1]
    g[key].append((d["value"], d.get("value_gram_cholesky")))
for k, v in sorted(g.items()):
    a = [x[0] for x in v]; b = [x[1] for x in v if x[1]]
    print("%-28s n=%d  mean %.1fk  min %.1fk  max %.1fk  %s" % (k, len(a), statistics.mean(a) / 1e3, min(a) / 1e3, max(a) / 1e3,
          ("| 2nd window mean %.1fk min %.1fk" % (statistics.mean(b) / 1e3, min(b) / 1e3)) if b else ""), [round(x / 1e3) for x in a])
