import sqlite3,sys
db=sqlite3.connect(sys.argv[1]); cur=db.cursor()
for r in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print("%-40s calls=%d total_us=%.1f avg_us=%.2f pct=%.2f"%(r[0].replace('(anonymous namespace)::','')[:40],r[1],r[2]/1e3 if r[2]>1e7 else r[2],r[3],r[4]))
