set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python -m pytest tests -m gpu -x -q -k "jump or fuzz or evalF or signature or greenlight" 2>&1 | tail -8
python tools/evalf_latency.py 300 > gpurun_out/r05/r05_evalf_latency.txt 2>&1; cat gpurun_out/r05/r05_evalf_latency.txt
GLGYM_TOOL_HEAVY=40 python tools/flag_tuples.py 400 2>&1 | tail -4
