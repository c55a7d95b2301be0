mkdir -p gpurun_out/r4t
for W in ${WAITERS:-32 256 1024}; do for H in ${HEAVIES:-2048}; do for P in ${PERIODS:-16}; do
  echo "== waiters $W heavy $H period $P"
  FMGPU_DEV_BOARD_WAITERS=$W FMGPU_DEV_BOARD_HEAVY=$H FMGPU_DEV_BOARD_PERIOD=$P PROBE_EDIT_ONLY=1 PROBE_REPEATS=1 FMGPU_DEV_BOARD_LOG=1 FMGPU_LIBRARY=fmindex-collection_amd/libfmgpu_dev.so timeout -k 10 200 python tools/batch_scaling_probe.py 2>&1 | grep -v amdgpu.ids | awk '/board:/{b=$0} /edit/{print $0; print "     ", b}'
done; done; done
