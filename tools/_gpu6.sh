set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for i in 1 2 3; do python tools/_diag_evalf5.py | grep -v "{(): 30}"; echo run $i; done
python -m pytest tests -m gpu -x -q -k "jump or fuzz or evalF or signature or greenlight or stress or storm" 2>&1 | tail -8
python tools/evalf_latency.py 300 > gpurun_out/r05/r05_evalf_latency.txt 2>&1; grep "1024\|#" gpurun_out/r05/r05_evalf_latency.txt
