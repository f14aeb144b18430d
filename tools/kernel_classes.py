import sys, json
for f in sys.argv[1:]:
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
    except Exception as e:
        print(f, "ERR", e); continue
    print(f, d['value'], d['ms_per_step'])
    for k, v in d.get('kernel_classes', {}).items():
        print(f"   {k:50s} {v['ms_per_step']:8.3f} ms  n={v['launches']:3d} tf={v.get('tflops')}")
