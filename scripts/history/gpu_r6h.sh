# round 5, last GPU call: the whole -m gpu suite and smoke() on the final HEAD (what the driver runs at round end)
D=gpurun_out/r6h
mkdir -p $D
export OMP_WAIT_POLICY=PASSIVE
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 > $D/suite.log 2>&1; echo "suite rc=$?"; tail -12 $D/suite.log
