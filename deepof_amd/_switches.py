"""Every environment switch the compiled library reads -- the complete list (tests/test_abi.py::test_switch_table_is_complete
compares it with the ``getenv`` sites of deepof_amd/csrc) -- and what each non-default value selects.  They exist for
same-box A/B measurements and for the parity tests, which run every non-default value through a reference check
(``probe``); a fit never needs one.  Each is read once per process.

Round 5 removed the switches whose alternatives no longer exist or no longer matter (DOF_GRU_MFMA, DOF_GRUM,
DOF_GRU8_MFMA, DOF_GRU8_FUSED, DOF_GRU8_FWD_PAIR, DOF_CONV_WGRAD_FUSED, DOF_TCN_ONEPASS, DOF_TCN_ONEPASS_MASK)."""

# name -> (default, tested non-default value, meaning of the non-default value, where it is exercised)
LIBRARY_SWITCHES = {
    "DOF_GRU_MFMA_MIN_S": ("8192", "0",
                           "sequences per launch from which the encoder GRU layers take the matrix-pipe kernels; 0 sends the small "
                           "reference goldens through them",
                           "tests/gru_mfma_probe.py (test_gru16_matrix_pipe_kernels_emu / _gpu)"),
    "DOF_GRU_WGRAD_FUSED": ("1", "0", "the lane-per-unit GRU layers (latent 4 - 7, 9, 10) write dG and the generic k_outer jobs reduce it, "
                            "instead of k_gru3_bwd accumulating the weight gradients on the matrix pipe",
                            "tests/gru_wgrad_probe.py (test_gru_unfused_weight_gradient_emu / _gpu)"),
    "DOF_OUTER_B3": ("1", "0", "the weight-gradient jobs on k_outer (v_mfma_f32_16x16x4_f32 on fp32 operands) instead of k_outer_b3 "
                     "(three-piece bf16 operands, v_mfma_f32_16x16x32_bf16)",
                     "tests/gru_wgrad_probe.py (test_outer_fp32_kernel_emu / _gpu)"),
    "DOF_TCN_WGRAD_FP32": ("0", "1", "TCN weight gradients on the fp32 k_outer reduction instead of the three-plane bf16 kernel",
                           "test_tcn_kernel_switches_gpu"),
    "DOF_TCN_TAIL_FOLD": ("1", "0", "the block tail's backward as its own launches instead of folded into the neighbouring convolution",
                          "test_tcn_kernel_switches_gpu"),
    "DOF_TCN_COMBINE_FOLD": ("1", "0", "the block output's forward as its own launch instead of folded into the next convolution",
                             "test_tcn_kernel_switches_gpu"),
    "DOF_TCN_STAT_RECORDS": ("1", "0", "BatchNorm batch statistics by a sum pass + a centred second pass instead of mergeable records",
                             "test_tcn_kernel_switches_gpu, test_tcn_record_statistics_vs_two_pass_gpu"),
    "DOF_TCN_WGRAD_IN": ("1", "0", "the first block's weight gradients on the staged kernel instead of the direct-load one",
                         "test_tcn_kernel_switches_gpu"),
    "DOF_TCN_CONV_B3": ("1", "0", "the time-resident TCN convolutions on round 5's fp32-MFMA kernels (k_tcn_conv_t, three workgroups per CU) "
                        "instead of the bf16-piece loader / compute kernel k_tcn_conv_b; also switches the fused weight gradient off",
                        "test_tcn_kernel_switches_gpu"),
    "DOF_TCN_WGRAD_FUSED": ("1", "0", "the 32-channel convolutions' weight gradients from k_tcn_wgrad_b3 (its own passes over dy, y and the "
                            "convolution input) instead of k_tcn_conv_b's data-gradient launches", "test_tcn_kernel_switches_gpu"),
    "DOF_TCN_LAST_BLOCK_SPARSE": ("1", "0", "the encoder's last block runs BatchNorm2's backward pass 1 over the whole window and writes its (zero) "
                                  "residual-branch gradient, instead of the last step alone + a never-written zero tensor",
                                  "test_tcn_kernel_switches_gpu"),
    "DOF_TCN_RESIDENT_MAX_T": ("50", "25", "longest window on the time-resident convolutions (longer windows take the 4-fetch path, "
                               "as windows > 50 always do)", "test_tcn_kernel_switches_gpu"),
}

# read by the Python host (deepof_amd.training / stepping), not by the library
HOST_SWITCHES = {
    "DOF_NO_GRAPH": "1: launch every step eagerly instead of replaying hipGraphs",
    "DOF_FORCE_DP": "1: take the data-parallel route with a 1-rank process group (tests / single-GPU measurements)",
    "DOF_DP_NATIVE": "0: torch.distributed.all_reduce between two graphs (the safe form) instead of dof_flat_allreduce",
    "DOF_DP_ONE_GRAPH": "0: two graphs around an eagerly enqueued collective; 1 with DOF_DP_NATIVE=0: capture the torch collective",
    "DOF_DP_SELF_CHECK": "0: skip the native collective's self-check (measurements only)",
    "DOF_DP_CHECK_TIMEOUT": "seconds the self-check waits for a collective (default 30)",
}
