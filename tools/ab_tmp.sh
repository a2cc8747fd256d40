export KB_T=250
for lib in libpfamd.so libpfamd_wt.so; do echo "== $lib"; PF_AMD_LIB=$PWD/pyfilter_amd/$lib python tools/kbench.py apf_lgo_1m apf_lgo_1024x8k apf_sv_64x64k apf_lgo_4m sisr_boot_1m 2>&1 | grep us/step; done
