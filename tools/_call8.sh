cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
timeout 900 python -m pytest tests/test_gpu_modules.py -x -q -k "vae or tryon_pipeline_tiny or graph" 2>&1 | tail -15
python - <<'PY'
import json;d=json.load(open('gpurun_out/parity_r03.json'));print(json.dumps(d.get('vae_decode_fp16_range_guard_tiny'),indent=1))
PY
