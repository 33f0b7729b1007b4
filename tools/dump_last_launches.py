import sqlite3, sys
db=sqlite3.connect(sys.argv[1]); n=int(sys.argv[2])
ev=sorted(db.execute("select start,end,name from kernels").fetchall())
names=[e[2].split("(")[0].replace("void ","").replace("cone::","") for e in ev]
sel=list(zip(ev[-n:],names[-n:])); t0=sel[0][0][0]; pe=t0
for (s,e,_),nm in sel:
    print(f"{(s-t0)/1e3:9.1f} +{(s-pe)/1e3:6.1f} {(e-s)/1e3:7.1f} {nm[:80]}"); pe=max(pe,e)
