# shader clock and power while a kernel family runs back to back (rocm-smi samples during a long probe loop)   usage: clock_under_load.sh <tag> <python script + args>
cd "$GRAFT_REPO_ROOT"; TAG=$1; shift; mkdir -p gpurun_out
( "$@" > gpurun_out/clock_${TAG}_run.log 2>&1 ) &
PID=$!
sleep ${WARM:-12}
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -iE "sclk|power" | tr -s ' ' | tr '\n' ';'; echo
  sleep 1
done | tee gpurun_out/clock_${TAG}.txt
wait $PID
tail -2 gpurun_out/clock_${TAG}_run.log
