cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
for sg in 0 -1 -2; do
  echo "probe $sg (0 = the kernel; -1 = no operand fetch inside the K loop; -2 = operands fetched but never waited for: timing probes, garbage results)"; WH_GEMM_PERSIST=1 WH_GEMM_STAGGER=$sg timeout 300 python tools/time_encoder.py large-v3 128 2>/dev/null | cut -c1-420
done
