cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
( time python bench.py ) > gpurun_out/r04_bench1.json 2> gpurun_out/r04_bench1.err
tail -5 gpurun_out/r04_bench1.err
python -m pytest tests/test_bench_multirank_gpu.py -q -x 2>&1 | tail -3
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r04_bench1.json') if x.startswith('{')]
d=json.loads(l[0])
print(d['value'], d['ms_per_step'], d['compress_MBps'], d['decompress_MBps'], d['roofline']['frac'], d['verified'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
for k,v in d['configs'].items():
    print(k, {kk:vv for kk,vv in v.items() if kk in ('ms_per_step','compress_MBps','decompress_MBps','compress_ms','decompress_ms','kernel_ms','verified')})
    cb=v.get('cpu_baseline'); print('   cpu', cb and {kk:cb[kk] for kk in cb if kk in ('value','cores','compress_MBps','decompress_MBps','t1')})
print(d.get('end_to_end'))
PY
