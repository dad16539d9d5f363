# round 6: the GPU suite at the final HEAD (after adaptive projected guidance / shifted_timestep), smoke
D=gpurun_out/r09last; mkdir -p $D
timeout 1800 python -m pytest tests -m gpu -x -q > $D/suite.log 2>&1; tail -3 $D/suite.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
