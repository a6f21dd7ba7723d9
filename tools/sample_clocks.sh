#!/bin/bash
# Sample the GPU's shader clock / power while bench.py's sustained leg runs: tools/sample_clocks.sh OUTDIR [seconds]
# (is the fp32-MFMA peak of 157.3 TFLOP/s -- 2.4 GHz x 256 CUs x 256 FLOP/clk -- the clock the stream actually runs at?)
OUT=$1; SECS=${2:-12}
mkdir -p $OUT
python bench.py --no-cpu-baseline --no-pcie --no-mgfn-train --steps 40 --warmup 6 --sustain-s $SECS > $OUT/bench.json 2> $OUT/bench.err &
BP=$!
sleep 2
for i in $(seq 1 $((SECS * 4 + 40))); do
  if ! kill -0 $BP 2>/dev/null; then break; fi
  rocm-smi --showclocks --showpower --showtemp --json 2>/dev/null | tr -d '\n' >> $OUT/smi.jsonl; echo >> $OUT/smi.jsonl
  sleep 0.5
done
wait $BP
echo "bench rc=$?"
python - <<PY
import json
rows=[json.loads(l) for l in open("$OUT/smi.jsonl") if l.strip().startswith("{")]
def pick(d, sub):
    for k, v in d.items():
        if sub in k.lower():
            return v
for r in rows[::4]:
    c = r.get("card0", {})
    print({k: v for k, v in c.items() if any(s in k.lower() for s in ("sclk", "mclk", "power", "temperature (sensor junction)", "fclk"))})
PY
python -c "import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['sustained'])"
