timeout -k 10 300 python tools/exp_epilogue_operands.py 2>&1 | tail -12
