for n in 0 3 4 5; do
  echo "== JCM_PERSIST_WGS=$n"
  JCM_PERSIST_WGS=$n python bench.py --dtype bf16 --steps 5 --warmup 2 --cpu-images 0 2> gpurun_out/occ_$n.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
  sort gpurun_out/occ_$n.err | uniq -c | grep persistent_grid | head -12
done
