for t in test_binary test_unary test_scale_and_timestep_embedding test_group_norm_chain test_layer_norm_chain test_soft_max test_linear_weight_gemm test_linear_residual_fusion_and_batch_dims test_generic_matmul_batched test_conv2d_chain test_conv2d_direct_and_residual test_layout_ops test_geglu_chain test_flash_attn_ext test_manual_attention_chain test_linear_split_k test_conv2d_split_k test_projection_head_major_chain test_feed_forward_geglu_fused; do
  r=$(timeout 120 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x -k "$t or (mmdit_forward and False)" 2>&1 | tail -1)
  echo "$t => $r"
done
