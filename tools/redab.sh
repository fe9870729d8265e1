# developer tool (GPU box): k_log_reduce16 by wavelengths per work item (DRT_LOG_LG = 2, 4, 8) on the bench step (kernel times from rocprofv3)
# usage: tools/redab.sh 4 8
export DRT_LOG=1 DRT_LOG_GB=100 TMPDIR=/tmp
for v in "$@"; do
  rm -rf gpurun_out/redab_kt
  DRT_LOG_LG=$v rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/redab_kt -o kt -- python3 bench.py --arena-gb 120 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/redab_bench.log 2>&1
  grep -o '"checksum_u_final_first_pass": [0-9.e+]*' gpurun_out/redab_bench.log | tail -1
  python3 - "$v" <<'PY'
import csv,sys,glob
f=glob.glob('gpurun_out/redab_kt/**/*kernel_trace.csv', recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
seq=[(r['Kernel_Name'], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6) for r in rows]
red=[ms for n,ms in seq if 'k_log_reduce' in n][-8:]
tr=[ms for n,ms in seq if 'rt_stage_wide' in n][-8:]
bn=[ms for n,ms in seq if 'k_log_count' in n or 'k_log_scatter' in n][-16:]
print('DRT_LOG_LG=%s'%sys.argv[1], 'last 8 reduce launches %.0f ms, trace %.0f ms, bin %.0f ms'%(sum(red), sum(tr), sum(bn)))
PY
done
