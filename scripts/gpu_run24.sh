for o in "" "fuse_gate=0" "fuse_modulate=0" "fuse_gelu=0" "fuse_gate=0,fuse_modulate=0,fuse_gelu=0"; do
  r=$(SDCPP_BACKEND_OPTS="$o" timeout 200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x -s 2>&1 | grep -E "SD35_TINY.*rel-L2|passed|failed" | tail -3 | tr '\n' ' ')
  echo "[$o] => $r"
done
