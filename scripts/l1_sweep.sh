for a in ${L1_SWEEP:-0 4 8}; do echo "ablate $a:"; TEDSPAD_L1_ABLATE=$a timeout -k 10 100 python scripts/bneck_l1_probe.py 2>&1 | grep "whole block\|diff"; done
