"""print a window of consecutive kernel dispatches (name, grid, duration) from a rocprofv3 *_kernel_trace.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
lo, n = int(sys.argv[2]), int(sys.argv[3])
for r in rows[lo:lo + n]:
    print('%-60s grid %7s  %6.1f us' % (r['Kernel_Name'].split('(')[0][-60:], r['Grid_Size_X'], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
