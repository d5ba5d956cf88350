#!/bin/bash
# development aid: gradient deviation of the one-pass statistics with single layers switched on
for m in ffffffff ffff00ff ffffff00 ff00ffff 00ffffff; do
  DOF_TCN_ONEPASS=1 DOF_TCN_ONEPASS_MASK=$m python tools/onepass_diag.py m$m > /dev/null 2>&1
done
DOF_TCN_ONEPASS=1 python tools/onepass_diag.py one > /dev/null 2>&1
