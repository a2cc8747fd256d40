export KB_T=250
for bi in 0 2; do echo "== PF_BOOK_INLINE=$bi"; PF_BOOK_INLINE=$bi python tools/kbench.py apf_lgo_1m sisr_boot_1m apf_lgo_4m 2>&1 | grep us/step | cut -c1-80; done
