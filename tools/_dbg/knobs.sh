run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/k_$name.json 2>gpurun_out/k_$name.err; }
run base A=1 && run pc48 LGCN_SWEEP_PIECE_CAP=48 && run pc112 LGCN_SWEEP_PIECE_CAP=112 && run la128 LGCN_SWEEP_LOOKAHEAD=128 && run la32 LGCN_SWEEP_LOOKAHEAD=32 && run base2 A=1
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/k_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["roofline"]["launch_ms"])
    except Exception as e: print(f, "ERR", e)
PY
