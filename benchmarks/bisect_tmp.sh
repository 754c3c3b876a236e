for e in "X=1" "REFTR_ADAMW_NT=0" "REFTR_PREP_SIDE=0" "REFTR_STREAMS=0"; do
  echo "== $e"
  env $e REPS=16 timeout 600 python benchmarks/debug_graph_vs_eager.py 2>&1 | grep "^[0-9]" | awk '{print $3}' | sort | uniq -c
done
