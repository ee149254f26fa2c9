"""per-(kernel, grid) durations of a rocprofv3 kernel_trace.csv: python3 tools_dev/trace_by_grid.py <csv>"""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:34]
    g = (int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))
    agg[(name, g, r['VGPR_Count'], r['LDS_Block_Size'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(agg.items()):
    print(f"{k[0]:34s} wgs={str(k[1]):18s} vgpr={k[2]:4s} lds={k[3]:6s} n={len(v):5d} avg_us={sum(v)/len(v):9.1f} min={min(v):9.1f} max={max(v):9.1f} total_ms={sum(v)/1e3:9.2f}")
