export KB_T=250
for lib in libpfamd.so libpfamd_w3.so; do echo "== $lib"; PF_AMD_LIB=$PWD/pyfilter_amd/$lib python tools/kbench.py sisr_lorenz_4m_mn sisr_lorenz_4m 2>&1 | grep us/step; done
