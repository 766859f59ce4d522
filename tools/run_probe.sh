#!/bin/bash
# GEMM probe over the shapes of the two headline configs (+ small / ragged ones)
set -x
cd "$(dirname "$0")/.."
P=tools/cgemm_probe
$P 2 64 256 512 512 3
$P 2 24 100 128 256 3
$P 2 24 5 64 9 3
$P 3 64 64 512 512 3
$P 3 24 100 128 256 3
$P 3 24 2 64 9 3
$P 3 40 64 512 9 3
$P 3 40 200 128 20 3
# full-size timing: conv5 and conv4_fullres of both configs (3136 frequencies), the logits layer
$P 2 3136 256 512 512 5
$P 2 3136 256 256 512 5
$P 3 3136 64 512 512 5
$P 3 3136 64 256 512 5
$P 3 3136 64 512 9 5
