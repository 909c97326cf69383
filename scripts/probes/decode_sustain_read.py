import csv, glob, os, sys
f = max(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = sorted((r for r in csv.DictReader(open(f)) if "decode_split" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
series, cur, last_end = [], [], None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if last_end is not None and s - last_end > 200e6 and cur:       # > 0.2 s idle: next phase
        series.append(cur); cur = []
    cur.append((e - s) / 1e3); last_end = e
series.append(cur)
for i, sr in enumerate(series):
    if len(sr) < 50: continue
    print(f"phase {i}: {len(sr)} launches; mean of each 20: " + " ".join(f"{sum(sr[j:j+20])/len(sr[j:j+20]):.1f}" for j in range(0, len(sr), 20)))
