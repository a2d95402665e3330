set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r03_m_nt_ab.txt
rm -f $O
timeout -k 5 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" 2>&1 | tail -5 | tee -a $O
timeout -k 5 300 python tools/nt_gemm_probe.py > gpurun_out/r03_nt_gemm_probe.txt 2>&1
one() {  # $1 = env assignment, $2.. = bench flags
  v=$1; shift
  echo "== $v  $*" >> $O
  env $v timeout -k 5 200 python bench.py --no-cpu-baseline --no-secondary --no-record --windows 1 --steps 30 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; k=r['kernels'].get('gemm_tt_dma128_table_kernel', {}); print('samples/s', d['value'], 'step ms', d['ms_per_step'], '| table launch us', k.get('avg_us'))" >> $O 2>&1
}
for v in "MTN_KEEP_WT=" "MTN_KEEP_WT=all" "MTN_KEEP_WT=qkv" "MTN_KEEP_WT=qkv,w1" "MTN_KEEP_WT=w1" "MTN_KEEP_WT=" "MTN_KEEP_WT=all"; do one $v; done
for v in "MTN_KEEP_WT=" "MTN_KEEP_WT=all" "MTN_KEEP_WT=qkv" "MTN_KEEP_WT=qkv,w1" "MTN_KEEP_WT=w1" "MTN_KEEP_WT=" "MTN_KEEP_WT=all"; do one $v --batch-per-gpu 64; done
cat $O
cd /tmp && export TMPDIR=/tmp
for v in "MTN_KEEP_WT="; do
  rm -rf /tmp/pl_stats
  env $v timeout -k 5 300 rocprofv3 --kernel-trace --stats -d /tmp/pl_stats -- python $R/bench.py --no-cpu-baseline --no-secondary --no-record --windows 0 --steps 10 > /tmp/pl_b.log 2>&1
  (cd $R && python tools/prof_breakdown.py /tmp/pl_stats 60 /tmp/seq_$v.txt > gpurun_out/r03_m_breakdown_none.txt)
done
