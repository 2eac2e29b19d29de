import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "ERR", e); continue
    p=d.get('passes',{}); q=d.get('passes_serial',{})
    print("%-40s %8.1f in flight (%.4f ms) | serial %8.1f | passes %s | serial %s" % (f.split('/')[-1], d['value'], d['ms_per_step'], d.get('value_serial',0), {k:round(v['ms_avg'],4) for k,v in p.items()}, {k:round(v['ms_avg'],4) for k,v in q.items()}))
