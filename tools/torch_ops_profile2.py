import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
exec(open(os.path.join(os.path.dirname(__file__), "torch_ops_profile.py")).read().split("print(prof.key_averages")[0])
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0 and ev.self_device_time_total <= 0:
        continue
    st = [s for s in (ev.stack or []) if "snn_for_object_detection_amd" in s or "tools/" in s]
    key = (ev.name, st[0].split("snn_for_object_detection_amd/")[-1][:70] if st else "(autograd engine / no python frame)")
    agg[key][0] += 1
    agg[key][1] += ev.self_device_time_total
for (name, where), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{name:28s} {n:4d} {t:9.1f} us  {where}")
print("total aten self device us:", sum(v[1] for v in agg.values()))
