#!/bin/bash
cd "$(dirname "$0")/../.."
python tools/fc_adam_probe.py planes3
python tools/fc_adam_probe.py exact
export SH_KERNEL_LIB=$PWD/semantichuman_amd/lib_alt/libsh_kernels.so
python tools/fc_adam_probe.py planes3
