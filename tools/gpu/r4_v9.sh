#!/bin/bash
# round 4 visit 9: reference-under-autocast figures of the full-size parity cases (BASELINE configs 2-5 at batch 2, the benchmark's
# batch 32 without conditioning) -> tests/golden/autocast_yardstick.json, then the batch-32 oracle comparison and the full-size tests
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v9; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
free -g | head -2
cp tests/golden/autocast_yardstick.json $D/autocast_yardstick.json
timeout 1500 python tools/autocast_yardstick.py --full --out $D/autocast_yardstick.json 2>&1 | grep -v Warning | tee $D/yardstick.log | cut -c1-400
cp $D/autocast_yardstick.json tests/golden/autocast_yardstick.json
rm -f $D/parity.jsonl
SF_PARITY_REPORT=$PWD/$D/parity.jsonl timeout 1500 python -m pytest -q -m gpu -x --tb=short tests/test_model_gpu.py -k "full_size" > $D/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $D/pytest.log | cut -c1-600
python - <<'PY'
import json
for l in open("gpurun_out/v9/parity.jsonl"):
    d = json.loads(l)
    y = d.get("reference_under_autocast") or {}
    print(d["case"], {k: round(d[k], 6) for k in ("logits_l2", "logits", "loss", "grad_norm", "grad_global", "grad_global_masked") if k in d},
          "| autocast:", {k: round(y[k], 6) for k in ("logits_l2", "logits", "loss", "grad_norm", "grad_global") if k in y})
PY
echo "exit 0"
