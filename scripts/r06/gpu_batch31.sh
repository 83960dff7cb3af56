#!/bin/bash
# complex add / subtract of the FFT engines as v_pk_add_f32 (common.hpp: DINV_PK_COMPLEX): parity tests and timings
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -q -m gpu -k "mri or fft or blur or ramp or tomo or radon" 2>&1 | tail -3
timeout 300 python scripts/r06/bench_mri_lanes.py 2>&1 | grep "lanes\": 1"
timeout 600 python scripts/r06/bench_blur.py 2>&1 | grep "BlurFFT\|MRI" | cut -c1-200
