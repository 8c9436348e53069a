import csv, sys, collections
def load(f):
    d=collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        d[r["Name"]]=(int(r["Calls"])/13.0, float(r["TotalDurationNs"])/13e6, float(r["AverageNs"])/1e3)
    return d
a,b=load(sys.argv[1]),load(sys.argv[2])
rows=[]
for k in set(a)|set(b):
    ta=a.get(k,(0,0,0)); tb=b.get(k,(0,0,0))
    rows.append((tb[1]-ta[1],k,ta,tb))
rows.sort(reverse=True)
print("total", sum(v[1] for v in a.values()), sum(v[1] for v in b.values()))
for d,k,ta,tb in rows[:22]: print("%+7.3f ms  %-90s A %5.1f x %7.2f us | B %5.1f x %7.2f us" % (d,k[:90],ta[0],ta[2],tb[0],tb[2]))
for d,k,ta,tb in rows[-6:]: print("%+7.3f ms  %-90s A %5.1f x %7.2f us | B %5.1f x %7.2f us" % (d,k[:90],ta[0],ta[2],tb[0],tb[2]))
