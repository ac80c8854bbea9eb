import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from oracle import ser_fast_path as ofp
from ser_amd import FeatureFlags
from ser_amd.engine import get_engine
from ser_amd.synth import harmonic_clips, sample_wav_like, ravdess_sine_clip
from parity_utils import block_errors
print("device", torch.cuda.get_device_name(0))
eng = get_engine()
flags = FeatureFlags(tonnetz=False)
sr = 16000
def run(name, audio, offs, lens):
    t0=time.time(); out, tun = eng.features(audio, offs, lens, sr, flags, return_tuning=True); torch.cuda.synchronize(); t1=time.time()
    got = out.cpu().numpy(); tun = tun.cpu().numpy()
    want=[]; wt=[]
    for o,l in zip(offs,lens):
        v,d = ofp.extract_feature_from_signal(audio[o:o+l], sr, return_debug=True); want.append(v); wt.append(d["tuning"])
    want=np.array(want); wt=np.array(wt)
    print(name, "gpu %.3fs"%(t1-t0), "errors", block_errors(got, want), "tuning gpu", tun[:6], "oracle", wt[:6], "tuning mismatches", int((np.abs(tun-wt)>1e-6).sum()), "of", len(wt))
    if not np.isfinite(got).all(): print("  NON-FINITE output!")
    return got, want
clips = harmonic_clips(1234, 8, 48000)
audio = clips.reshape(-1)
run("8x3s", audio, [i*48000 for i in range(8)], [48000]*8)
a = sample_wav_like()
offs = list(range(0, a.size, 16000)); lens=[min(48000, a.size-o) for o in offs]
run("samplewav", a, offs, lens)
rav = np.concatenate([ofp.prepare_audio_buffer(ravdess_sine_clip(ac, e)) for ac in (1,2) for e in range(1,9)])
run("ravdess", rav, [i*24000 for i in range(16)], [24000]*16)
rng = np.random.default_rng(5)
x = rng.standard_normal(30000).astype(np.float32); x/=np.abs(x).max()
run("ragged", x, [0, 1, 4001, 100, 7], [2048, 2049, 5937, 20000, 29993])
run("short", x, [0, 10, 33, 500, 77], [100, 512, 1023, 1500, 2047])
z = np.zeros(6000, np.float32)
run("silence", z, [0], [6000])
