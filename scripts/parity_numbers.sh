#!/bin/bash
# The measured maxima the parity tests print (run with -s), for profiles/<tag>_parity_numbers.txt:
#   bash scripts/parity_numbers.sh r03_final
TAG=${1:-rXX}
OUT=gpurun_out/${TAG}_parity_numbers.txt
python3 -m pytest -m gpu -q -s -p no:cacheprovider \
  "tests/test_gpu_ops.py::test_split_bf16_convolutions_keep_fp32_accuracy" \
  "tests/test_gpu_ops.py::test_split_bf16_conv1_error_bound_per_element" \
  "tests/test_gpu_distributed.py::test_sharded_trajectory_left_alone_stays_within_the_drift_bound" \
  "tests/test_gpu_step.py::test_b5_1to8_single_gpu_runs_on_with_modulo_bank_writes" \
  "tests/test_gpu_step.py::test_device_noise_is_the_documented_generator" \
  "tests/test_gpu_step.py::test_first_step_gradients_with_the_oracles_own_relu_decisions" 2>&1 | grep -av amdgpu.ids > $OUT
tail -3 $OUT
