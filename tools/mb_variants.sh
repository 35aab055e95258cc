for v in 0 1 2 3 4; do echo "variant $v"; WT_ATTN_VARIANT=$v timeout -k 10 120 python tools/microbench.py dec_attn 2>&1 | grep -E "S=1500 n_split=(1|2|4)|len=447.*n_split=1"; done
