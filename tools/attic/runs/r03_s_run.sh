set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r03_k512_probe.txt
rm -f $O
for lib in "" k_STAMPS; do
  echo "== library: ${lib:-product}" >> $O
  if [ -n "$lib" ]; then export MTN_HIP_LIB=$R/tools/libmtn_hip_$lib.so; else unset MTN_HIP_LIB; fi
  timeout -k 5 120 python tools/k512_probe.py 2>&1 | grep -v amdgpu.ids >> $O
done
cat $O
