import sqlite3, sys
for f in sys.argv[1:]:
    cur=sqlite3.connect(f).cursor()
    rows=list(cur.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"))
    ks={}
    for k,c,n,s in rows:
        if 'solve' in k: ks.setdefault(k.split('(')[0],{})[c]=(n,s)
    for k,v in ks.items():
        print(k)
        w = v.get('SQ_WAVES',(0,0))[1]
        for c,(n,s) in sorted(v.items()): print(f"   {c:24s} n={n} sum={s:.4g}" + (f"  per wave {s/w:.1f}" if w else ""))
