"""ms_per_step and the level-0 mean-field figures of the A/B runs under gpurun_out/ab (scratch/run_ab.sh)."""
import glob, json
for f in sorted(glob.glob('gpurun_out/ab/*.json')):
    try:
        r = json.load(open(f))
    except Exception:
        continue
    rb = r.get('roofline_bwd') or {}
    print('%-52s step %.4f ms  mf fwd %.2f us  bwd %.2f us  layer fwd %s bwd %s' % (f.split('/')[-1], r['ms_per_step'], r['roofline']['avg_launch_us'],
          rb.get('avg_launch_us', 0), (r.get('roofline_layer') or {}).get('fwd_us'), (r.get('roofline_layer') or {}).get('bwd_us')))
