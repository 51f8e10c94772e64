mkdir -p gpurun_out/r06b
leg() { # name, args...
  name=$1; shift
  HRX_PLACE_TRACE=1 timeout 500 python bench.py "$@" --no-cpu-baseline --no-pmc > gpurun_out/r06b/$name.json 2> gpurun_out/r06b/$name.err
  python - <<PY
import json
l=json.loads(open("gpurun_out/r06b/$name.json").read().strip().splitlines()[-1]); r=l["roofline"]
print("$name frac", round(r["frac"],4), "per_set", r.get("per_set_ms"), "interleaved", (r.get("interleaved_layout") or {}).get("frac"), r["placement"].get("search_ms"))
PY
  grep -h "every pairing collides\|round [1-4]," gpurun_out/r06b/$name.err | head -12 | cut -c1-200
}
C5="--config dfa256 --batch 131072 --len 4095 --rows 4096 --steps 10 --warmup 3 --distinct 65536"
C4="--config headers3 --batch 32768 --len 32767 --rows 32768 --steps 5 --warmup 2 --distinct 4096"
C3="--config regex23 --batch 1048576 --len 2047 --rows 2048 --steps 5 --warmup 2 --distinct 65536"
leg c5_pair $C5
leg c5_batch $C5 --planes
leg c5_batch2 $C5 --planes
leg c5_batch3 $C5 --planes
leg c5_pair2 $C5
leg c4_batch $C4 --planes
leg c3_batch $C3 --planes
