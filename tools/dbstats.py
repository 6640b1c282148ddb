import sqlite3, sys, csv
db=sqlite3.connect(sys.argv[1]); cur=db.cursor()
rows=list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
n=int(sys.argv[2]) if len(sys.argv)>2 else 7
tot=sum(r[2] for r in rows)
print(f"total kernel time {tot/1000:.2f} ms  -> {tot/1000/n:.2f} ms/utt over {n} utts")
for r in rows[:int(sys.argv[3]) if len(sys.argv)>3 else 22]:
    print(f"{r[4]:6.2f}%  calls/utt {r[1]/n:7.1f}  avg {r[3]:8.2f} us  tot/utt {r[2]/n/1000:7.3f} ms  {r[0][:90]}")
if len(sys.argv)>4:
    with open(sys.argv[4],'w',newline='') as f:
        w=csv.writer(f); w.writerow(['Name','Calls','TotalDurationUs','AverageUs','Percentage'])
        for r in rows: w.writerow(r)
