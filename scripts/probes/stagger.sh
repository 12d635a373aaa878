for f in 0 4 8 12 16 20; do echo "== stagger fwd $f"; CMLPL_STAGGER_F=$f bash scripts/kstats.sh ks_$f B2 2>&1 | grep "conv3x3_kernel<2\|sum per"; done
for b in 4 8 12 16; do echo "== stagger bwd $b"; CMLPL_STAGGER_B=$b bash scripts/kstats.sh ksb_$b B2 2>&1 | grep "conv3x3_kernel<3\|sum per"; done
