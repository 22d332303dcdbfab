"""print the launches >= MIN us of one step of a tools/pmc_summary.py timeline, starting at the first kernel whose name contains KEY"""
import sys
path, key, mn = sys.argv[1], sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else 9.0
rows = [l.rstrip("\n") for l in open(path) if not l.startswith("#")]
idx = [i for i, l in enumerate(rows) if key in l]
start = idx[-1] if idx else 0
tot = {}
for l in rows[start:]:
    f = l.split(None, 4)
    d = float(f[1])
    name = f[4].split("<")[0]
    tot[name] = tot.get(name, 0.0) + d
    if d >= mn:
        print(l[:130])
print("---- per kernel family (us) from", key)
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:25]:
    print("%9.1f  %s" % (v, k))
print("%9.1f  total" % sum(tot.values()))
