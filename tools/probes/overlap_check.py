"""do consecutive step launches overlap in time?  (rocprofv3 kernel trace of a few train steps: start/end of the k_mlp_chain kernels)"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'k_mlp_chain' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
lo = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
t0 = int(rows[lo]['Start_Timestamp'])
for r in rows[lo:lo + 14]:
    print('%-22s queue %s stream %s  start %8.1f us  end %8.1f us' % (r['Kernel_Name'].split('(')[0][-22:], r['Queue_Id'], r['Stream_Id'],
          (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3))
