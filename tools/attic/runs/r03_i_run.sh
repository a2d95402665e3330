set -x
R=$GRAFT_REPO_ROOT
cd $R
rm -f gpurun_out/r03_i_corpus.txt
for v in "X=1" "MTN_FB_MAX_A=32" "X=1" "MTN_FB_MAX_A=32"; do
  echo "== ragged corpus (answers <= 52, questions <= 42 tokens), batch 32, 3 epochs: $v" >> gpurun_out/r03_i_corpus.txt
  env $v timeout -k 5 280 python -m mtn_amd.train --corpus-videos 48 --corpus-max-answer 52 --corpus-max-question 42 --num-epochs 3 --batch-size 32 --report-interval 5 2>&1 | grep -E "Tokens per Sec|epoch" | tail -6 >> gpurun_out/r03_i_corpus.txt
done
cat gpurun_out/r03_i_corpus.txt
