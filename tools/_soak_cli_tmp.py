import json, os, subprocess, sys, tempfile
ROOT = os.getcwd()
CFG = "cf_IAMslant_noMask_charSpecSingleAppend_GANMedMT_autoAEMoPrcp2tightNewCTCUseGen_balB_hCF0.75_sMG.json"
tmp = tempfile.mkdtemp()
cfg = json.load(open(os.path.join(ROOT, "configs", CFG)))
tr = cfg["trainer"]
tr.update(save_dir=tmp + "/saved", save_step=140, save_step_minor=70, log_step=35, val_step=0, print_dir=None, async_log=2,
          encoder_weights=tmp + "/enc/encoder.pth", text_data=tmp + "/no_corpus.txt")
cfg["model"]["pretrained_hwr"] = None
path = tmp + "/" + CFG
json.dump(cfg, open(path, "w"))
r = subprocess.run([sys.executable, os.path.join(ROOT, "train.py"), "-c", path, "--synthetic", "--iterations", "420"], cwd=tmp, env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True)
print(r.returncode); print((r.stdout + r.stderr)[-1500:])
print(sorted(os.listdir(os.path.join(tmp, "saved", cfg["name"]))))
