import sqlite3, sys
db=sqlite3.connect(sys.argv[1])
rows=db.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection group by kernel_name, counter_name").fetchall()
d={}
for k,c,n,v,dur in rows:
    short=k.split("(")[0].replace("(anonymous namespace)::","").replace("void ","")[:48]
    d.setdefault(short,{})[c]=v; d[short]["_dur_us"]=dur/1e3; d[short]["_n"]=n
for k,v in sorted(d.items(), key=lambda kv:-kv[1]["_dur_us"]*kv[1]["_n"])[:int(sys.argv[2])]:
    print(k, {a:(round(b,1) if b<1e4 else f"{b:.3g}") for a,b in v.items()})
