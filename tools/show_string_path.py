import json,sys
for f in sys.argv[1:]:
    d=json.loads(open(f).read().strip().splitlines()[-1])
    def find(o,k):
        if isinstance(o,dict):
            if k in o: return o[k]
            for v in o.values():
                r=find(v,k)
                if r is not None: return r
        return None
    sp=find(d,'string_path')
    print(f)
    for m,row in sp['sizes'].items():
        print(' ',m,{k:(round(v['decisions_per_s']/1e6,1),round(v['p50_ms'],3)) for k,v in row.items()})
