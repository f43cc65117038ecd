"""Barrier-delimited stages of yoloface56_fused, in stop_stage order (stop_stage k ends the kernel after stage k)."""
NAMES = ["input staging", "conv2d_1", "conv2d_3 (dw)", "conv2d_5", "conv2d_6", "pool_8 h", "pool_8 v + conv2d_10 (dw)", "conv2d_12",
         "conv2d_13", "conv2d_15 (dw)", "conv2d_17+add", "conv2d_19", "conv2d_23", "pool_25 + conv2d_27 (dw)", "conv2d_29",
         "conv2d_30", "conv2d_32 (dw)", "conv2d_34+add", "conv2d_36", "conv2d_38 (dw)", "conv2d_40+add", "conv2d_42", "conv2d_47",
         "conv2d_49 (dw)", "conv2d_51", "conv2d_53 + store"]
