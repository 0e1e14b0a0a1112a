import cProfile, pstats, sys, os, io
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_)
import torch  # noqa
from dbat_amd import bundle, synth, _hip
s, _ = synth.make_scene(sys.argv[1] if len(sys.argv) > 1 else 'C3')
bundle(s, 'lm', store_trace=False)
pr = cProfile.Profile(); pr.enable()
r = bundle(s, 'lm', store_trace=False)
pr.disable()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats('cumulative').print_stats(28); print(st.getvalue()[:6000])
print(r[4].timeHost)
