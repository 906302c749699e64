"""Where the Gram-variant step spends its time: front (weight-independent forward), rest of the forward up to the
point where the eigen jobs are issued, the eigen wait, backward + update.  HIP events on the main stream."""
import sys, os, warnings, contextlib, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_ops as R
from style_transfer_amd.trainer import GradBucket, Trainer
dev = torch.device("cuda")
with warnings.catch_warnings(), contextlib.redirect_stdout(sys.stderr):
    warnings.simplefilter("ignore")
    tr = Trainer({"train": []}, output_dir="/tmp/st", num_train_epochs=1, per_device_batch_size=8, learning_rate=1e-4,
                 alpha=1.0, beta=10.0, gamma=1.0, delta=10.0, eps=0.1, do_decoder_train=True,
                 optim_name={"optim_name": "adam"}, use_VincentStyleLoss=True, content_layers_idx=[21],
                 style_layers_idx=[2, 7, 14, 19, 25], encoder_weights=R.make_encoder_state(0),
                 vgg_weights=R.make_loss_vgg_state(2, upto=25), device=dev)
    enc, dec, ada, vgg = tr.build_model()
    dec.load_state_dict(R.make_decoder_state(1))
    params = tr.trainable_parameters(ada, dec)
    opt = tr.get_optimizer("adam", params, 1e-4, {})
    step = tr.make_step(enc, dec, ada, vgg)
bucket = GradBucket(params)
g = torch.Generator().manual_seed(1)
c = torch.rand(8, 3, 256, 256, generator=g).to(dev); s = torch.rand(8, 3, 256, 256, generator=g).to(dev)
for _ in range(4):
    tr.optimize_step(step, opt, bucket, c, s)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
ev[0].record()
fr = step.front(c, s)
ev[1].record()
bucket.zero()
marks = {}
def hook():
    ev[2].record()
partials = step.rest(fr, hook)
ev[3].record()
losses = tr.step_backward(step, partials, partials[3], 8)
ev[4].record()
tr.step_update(step, opt, bucket)
ev[5].record()
torch.cuda.synchronize()
names = ["front", "rest of forward up to eigen jobs issued", "eigen wait + loss tail", "backward", "update"]
for i, n in enumerate(names):
    print(f"{n:45s} {ev[i].elapsed_time(ev[i + 1]):7.2f} ms")
print(f"{'total':45s} {ev[0].elapsed_time(ev[5]):7.2f} ms")
