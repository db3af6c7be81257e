"""CPU: which convolutions the GAN's bf16-storage mode routes to the bf16 forms (sequitr_amd/ops_gan_bf16.takes) -- the rule is
by dtype and shape only, so it is checked without a GPU."""
import torch

from sequitr_amd import ops
from sequitr_amd import ops_gan_bf16 as gb


def test_store_flag_follows_the_precision_context():
    assert not ops.STORE_BF16 and not ops.MIXED
    with ops.mixed_precision(True, store_bf16=True):
        assert ops.STORE_BF16 and ops.MIXED
        with ops.mixed_precision(True):                        # the 'mixed' form nested inside: f32 storage
            assert ops.MIXED and not ops.STORE_BF16
        assert ops.STORE_BF16
    assert not ops.STORE_BF16 and not ops.MIXED
    with ops.mixed_precision(False, store_bf16=True):          # storage never without the bf16 multiplies
        assert not ops.STORE_BF16


def test_only_images_enter_bf16_storage():
    img = torch.zeros((4, 16, 16, 2))
    w_from = torch.zeros((1, 1, 2, 32))
    w_to = torch.zeros((1, 1, 32, 2))
    logits_grad = torch.zeros((1, 1, 4, 1))                     # the (1,1,N,1) row form of the discriminator's logits gradient
    w_logits = torch.zeros((1, 1, 32, 1))
    assert not gb.takes(img, w_from)                            # flag off: the f32 / mixed graph
    with ops.mixed_precision(True, store_bf16=True):
        assert gb.takes(img, w_from)                            # from_image: image -> features
        assert gb.takes(img, w_to, dgrad=True)                  # gradient of to_image's input: image gradient -> features
        assert not gb.takes(logits_grad, w_logits, dgrad=True)  # a dense layer's row form is not an image
        assert not gb.takes(torch.zeros((4, 16, 16, 16)), torch.zeros((3, 3, 16, 16)))   # f32 features stay f32 (mixed conv)
        assert not gb.takes(img, torch.zeros((3, 3, 2, 32)))    # image-side convs are 1x1
    assert gb.takes(torch.zeros((4, 8, 8, 16), dtype=torch.bfloat16), torch.zeros((3, 3, 16, 16)))   # bf16 in: always
