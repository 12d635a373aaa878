# Record of an experiment (profiles/r05_stagger_rounds.txt).  The CMLPL_STAGGER switch it drives is NOT in the library any more:
# workgroups with linear index in [CUs, 2 CUs) -- the CUs' second slots of the first round -- slept `stagger` x s_sleep(32) at
# kernel entry, set by the three per-sample launch functions when the grid exceeded two workgroups per CU.
# one-time stagger of the second-slot workgroups in multi-round launches: cube inference (405 rounds) and configs[2] whole on one GPU (4 rounds)
for d in 0 10 15 20 25 30 0; do echo "== CMLPL_STAGGER=$d"; CMLPL_STAGGER=$d python3 scripts/bench_infer.py B2 2>&1 | grep "207400 pixels per launch\|labels equal" ; CMLPL_STAGGER=$d python3 bench.py --workload B3 --gpus 1 --steps 100 --no-cpu-baseline 2> /dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('B3 512+512 on one GPU %.4f ms/step' % d['ms_per_step'])"; done
