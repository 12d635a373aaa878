# Record of an experiment (profiles/r05_stagger.txt).  The CMLPL_STAGGER_F / _B switches it drives are NOT in the library any more:
# they were three lines at the top of conv3x3_kernel -- `if (MODE >= 2 && a.stagger > 0 && blockIdx.y == 1) for (i < a.stagger) s_sleep(32);`
# -- with a.stagger set from the switch in launch_conv3_fused / launch_conv3_fused_bwd.
for f in 0 4 8 12 16 20; do echo "== stagger fwd $f"; CMLPL_STAGGER_F=$f bash scripts/kstats.sh ks_$f B2 2>&1 | grep "conv3x3_kernel<2\|sum per"; done
for b in 4 8 12 16; do echo "== stagger bwd $b"; CMLPL_STAGGER_B=$b bash scripts/kstats.sh ksb_$b B2 2>&1 | grep "conv3x3_kernel<3\|sum per"; done
