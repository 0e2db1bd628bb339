# HiFi-GAN iteration: bias gradients of the generator's resblock convolutions inside the weight-gradient launches (XVA_HG_BIAS_FUSED) on / off, same box
R=$GRAFT_REPO_ROOT; cd $R
for m in 0 1 0 1 0 1; do XVA_HG_BIAS_FUSED=$m python tools/hg_phase_timing.py 2>/dev/null | grep -E "gen_bwd|total" | tr '\n' ' '; echo " BIAS_FUSED=$m"; done
