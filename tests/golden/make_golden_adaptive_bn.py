"""Generates tests/golden/adaptive_bn_*.npz by running the imported reference (read-only, /root/reference) on the CPU.

Run once in the build container:  python tests/golden/make_golden_adaptive_bn.py
The forward is the one Training.adaptive_BN performs (filter_pruning/fp_train.py:231-240): model.train(), torch.no_grad(),
`imgs = inputs.float() / 255.0; model(imgs)` per uint8 batch of the loader.  The models are the reference's own
(models/yolo.py Model on its yaml files, seed 0, BatchNorm parameters and statistics randomised as for the end-to-end
fixtures, so that the weights can be rebuilt from the seed).  Stored: the uint8 batches and every BatchNorm's
running_mean / running_var afterwards, plus num_batches_tracked of the first one.  Data only, no source text.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg  # noqa: E402


def main():
    common, yolo, uu, tu = mg.import_reference()
    for cfg, shapes in (('yolov5s', ((8, 3, 128, 96), (8, 3, 96, 128))), ('yolov5s_ori', ((8, 3, 96, 96), (8, 3, 128, 64)))):
        torch.manual_seed(0)
        model = mg.quiet(yolo.Model, 'models/%s.yaml' % cfg)
        mg.randomize_bn(model, 11)
        model.train()
        g = torch.Generator().manual_seed(21)
        batches = [torch.randint(0, 256, s, generator=g, dtype=torch.uint8) for s in shapes]
        with torch.no_grad():
            for inputs in batches:
                imgs = inputs.float() / 255.0
                model(imgs)
        sd = model.state_dict()
        out = {'x%d' % i: b for i, b in enumerate(batches)}
        out.update({k: v for k, v in sd.items() if k.endswith('running_mean') or k.endswith('running_var')})
        out['num_batches_tracked'] = next(v for k, v in sd.items() if k.endswith('num_batches_tracked'))
        mg.save('adaptive_bn_' + cfg, **out)


if __name__ == '__main__':
    main()
