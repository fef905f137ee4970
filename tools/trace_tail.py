"""reduce a rocprofv3 kernel_trace.csv to the last N dispatches: start_us,end_us,queue,kernel"""
import csv, re, sys
src, dst, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
rows = list(csv.DictReader(open(src)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-n:]
t0 = int(tail[0]["Start_Timestamp"])
with open(dst, "w") as out:
    for r in tail:
        name = r["Kernel_Name"]
        m = re.search(r"(k_[a-z_0-9]+)", name)
        short = m.group(1) if m else name.split("::")[-1][:40]
        out.write("%.1f,%.1f,%s,%s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
                                         r.get("Queue_Id", ""), short))
