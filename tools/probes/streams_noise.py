"""Run-to-run spread of the first step's gradients with and without discriminator streams.
python tools/probes/streams_noise.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import warnings
warnings.filterwarnings('ignore')
import vcagan_amd  # noqa
from test_streams_gpu import _rel, _run

for prec in ('fp32', 'mixed'):
    runs = [(_run(s, prec, False)) for s in (False, False, True, True, True)]
    base = runs[0]
    for name, r in zip(('one-stream again', 'streams', 'streams again', 'streams third'), runs[1:]):
        print(prec, '%-18s' % name, 'G grads rel %.2e  D grads rel %.2e  dis_loss diff %.2e gen_loss diff %.2e' % (
            _rel(r[1]['g'], base[1]['g']), _rel(r[1]['d'], base[1]['d']),
            abs(r[0][0]['dis_loss'] - base[0][0]['dis_loss']), abs(r[0][0]['gen_loss'] - base[0][0]['gen_loss'])), flush=True)
