run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/m_$name.json 2>gpurun_out/m_$name.err; }
run base A=1 && run mix LGCN_TILE_ORDER=cold_mix && run base2 A=1 && run mix2 LGCN_TILE_ORDER=cold_mix && run nat LGCN_TILE_ORDER=natural
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/m_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["roofline"]["launch_ms"])
    except Exception as e: print(f, "ERR", e)
PY
