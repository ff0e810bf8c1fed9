import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if isinstance(v, dict) and "ms_per_mtu_call" in v:
        print("%-28s %7.1f us" % (k, 1e3 * v["ms_per_mtu_call"]))
