export KB_T=250
python tools/kbench.py apf_lgo_1m sisr_boot_1m apf_lgo_1024x8k apf_sv_64x64k apf_lgo_64x64k apf_lgo_4m sisr_lorenz_4m sisr_lorenz_4m_mn 2>&1 | grep us/step | cut -c1-80
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
