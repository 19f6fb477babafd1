import sys, struct
a=open(sys.argv[1]).read().splitlines(); b=open(sys.argv[2]).read().splitlines()
n=0
for i,(x,y) in enumerate(zip(a,b)):
    if x!=y:
        xs=x.split(); ys=y.split()
        print("line",i, xs[:2], "prev:", a[i-1][:110])
        for j,(u,v) in enumerate(zip(xs,ys)):
            if u!=v:
                try:
                    fu=struct.unpack('f',struct.pack('I',int(u,16)))[0]; fv=struct.unpack('f',struct.pack('I',int(v,16)))[0]
                    print("   tok",j,u,v,fu,fv)
                except Exception: print("   tok",j,u,v)
        n+=1
        if n>=int(sys.argv[3]) : break
print("lines",len(a),len(b))
