run() { name=$1; shift; env "$@" timeout -k 10 120 python tools/bench_encode.py --steps 10 ${COMPUTE} > gpurun_out/enc_$name.log 2>&1 || { tail -3 gpurun_out/enc_$name.log; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/enc_$name.log").read().strip().splitlines()[-1])
print("$name device_ms", round(d["device_ms"],3), "TF/s", round(d["roofline"]["achieved"],1))
PY
}
COMPUTE="--compute f32" run f32 X=1
COMPUTE="--compute bf16x3" run x3 X=1
COMPUTE="--compute bf16x3" run x3_ws PCV_GEMM_WS=1
COMPUTE="--compute f32 --ragged" run f32_ragged X=1
