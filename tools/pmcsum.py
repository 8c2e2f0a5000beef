"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, per-dispatch averages."""
import csv, collections, glob, re, sys
def short(n):
    m = re.search(r'(k_\w+)(<[^>]*>)?', n)
    return (m.group(1) + (m.group(2) or '')) if m else n[:30]
for d in sys.argv[1:]:
    for f in glob.glob(d + '/*/*_counter_collection.csv'):
        rows = list(csv.DictReader(open(f)))
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
        dur = collections.defaultdict(float); info = {}
        for r in rows:
            k = short(r['Kernel_Name'])
            if not k.startswith('k_'): continue
            agg[k][r['Counter_Name']] += float(r['Counter_Value'])
            if r['Dispatch_Id'] not in disp[k]:
                disp[k].add(r['Dispatch_Id']); dur[k] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
                info[k] = 'grid %s wg %s lds %s scratch %s vgpr %s agpr %s sgpr %s' % (r['Grid_Size'], r['Workgroup_Size'], r['LDS_Block_Size'], r['Scratch_Size'], r['VGPR_Count'], r['Accum_VGPR_Count'], r['SGPR_Count'])
        for k, v in agg.items():
            n = len(disp[k])
            if dur[k] / n < 2e5: continue
            print('%s  %s: %d dispatches, avg %.3f ms  [%s]' % (d, k, n, dur[k] / n / 1e6, info[k]))
            for c, val in sorted(v.items()): print('   %-28s %.4g' % (c, val / n))
