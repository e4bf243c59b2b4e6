"""Golden vectors of the composition stage from the REFERENCE's own module (build container only).

    python -m oracle.ref_harness.make_composition_golden

Imports /root/reference/core/UDIS2/Composition/network.py (plain torch, CPU), loads
``oracle.composition.seeded_state_dict(4321)`` strictly, runs ``build_model`` on ``oracle.composition.synthetic_inputs(512, 544)`` (regenerated
from the seed by the tests; an input checksum is stored) and stores the expected outputs.  Only data is written."""
import os
import sys

import numpy as np
import torch

from oracle import composition as oc

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden")


def main():
    sys.path.insert(0, "/root/reference")
    from core.UDIS2.Composition.network import Network, build_model
    torch.manual_seed(0)
    net = Network()
    sd = oc.seeded_state_dict(4321)
    assert list(net.state_dict().keys()) == list(sd.keys())
    net.load_state_dict(sd, strict=True)
    net.eval()
    out1, out2, m1, m2 = oc.synthetic_inputs(512, 544, 77)
    w1, w2 = oc.preprocess(out1, False), oc.preprocess(out2, False)          # out.py:284-286 (>= 512: no resize)
    with torch.no_grad():
        o = build_model(net, w1, w2, m1, m2)
        mask = net(w1, w2, m1, m2)
    np.savez_compressed(os.path.join(OUT, "composition_512x544.npz"),
                        in_checksum=np.array([float(out1.double().sum()), float(out2.double().sum()), float(m1.sum()), float(m2.sum())]),
                        net_out_sub=mask[0, 0, ::2, ::2].numpy().astype(np.float32),
                        stitched_sub=o["stitched_image"][0, :, ::4, ::4].numpy(), lm1_sub=o["learned_mask1"][0, :, ::4, ::4].numpy(),
                        lm2_sub=o["learned_mask2"][0, :, ::4, ::4].numpy(), keys=np.array(list(sd.keys())))
    print("net_out range", float(mask.min()), float(mask.max()), "stitched mean", float(o["stitched_image"].mean()))


if __name__ == "__main__":
    main()
