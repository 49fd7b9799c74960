"""GPU parity of the on-device CLIP text encoder (diffusion_pullback_amd/text_encoder.py) against the CPU oracle
(oracle/clip_text.py; PARITY UNPINNED: transformers' CLIPTextModel weights / tokenizer are not available offline),
row f4 of SURVEY.md section 8 (``pipe._encode_prompt``, reference src/modules/edit.py:505-522)."""
import pytest
import torch

from _util import rel

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.bfloat16, 3e-2)])
def test_small_clip_text_encoder(dtype, tol):
    from diffusion_pullback_amd import configs as cf
    from diffusion_pullback_amd.text_encoder import ClipTextEncoder
    from oracle import clip_text as oc
    cfg = cf.CLIPTextConfig(vocab_size=200, hidden=64, layers=3, heads=4, intermediate=128, max_position=24)
    p = cf.clip_init_params(cfg, seed=4)
    enc = ClipTextEncoder(cfg, p, dtype=dtype, device="cuda:0", max_batch=2)
    ids = torch.randint(0, cfg.vocab_size, (3, cfg.max_position), generator=torch.Generator().manual_seed(0))    # batch 3 > max_batch: chunking
    y = enc(ids).cpu()
    ref = oc.encode(p if dtype == torch.float32 else {k: v.to(dtype).float() for k, v in p.items()}, cfg, ids.long())
    assert y.shape == (3, cfg.max_position, cfg.hidden)
    assert rel(y, ref) < tol, rel(y, ref)
    # causality: tokens after position t do not change the outputs up to t
    ids2 = ids.clone(); ids2[:, 10:] = (ids2[:, 10:] + 7) % cfg.vocab_size
    y2 = enc(ids2).cpu()
    assert rel(y2[:, :10], y[:, :10]) < (1e-5 if dtype == torch.float32 else 1e-2)
    assert rel(y2[:, 10:], y[:, 10:]) > 1e-2
    with pytest.raises(ValueError):
        enc(ids[:, :5])
    with pytest.raises(RuntimeError):
        enc.encode_prompt("a photo of a dog")              # no tokenizer injected (vocabulary files are not available offline)


def test_sd15_clip_text_encoder_full_size_bf16():
    """The real CLIP ViT-L/14 text model shapes (49408 x 768 embedding, 12 layers, 12 heads of 64, 77 tokens)."""
    from diffusion_pullback_amd import configs as cf
    from diffusion_pullback_amd.text_encoder import ClipTextEncoder
    from oracle import clip_text as oc
    cfg = cf.SD15_CLIP
    p = cf.clip_init_params(cfg, seed=5)
    toks = lambda s: ([49406] + [1000 + (ord(c) % 500) for c in s][:75] + [49407] * 77)[:77]        # stand-in tokenizer: BOS, ids, EOS padding
    enc = ClipTextEncoder(cfg, p, dtype=torch.bfloat16, device="cuda:0", max_batch=1, tokenizer=toks)
    e = enc.encode_prompt("sitting dog").cpu()
    ids = torch.tensor([toks("sitting dog")])
    ref = oc.encode({k: v.to(torch.bfloat16).float() for k, v in p.items()}, cfg, ids)
    assert e.shape == (1, 77, 768) and torch.isfinite(e).all()
    assert rel(e, ref) < 3e-2, rel(e, ref)


def test_sd21_openclip_text_encoder_shapes_fp16_and_bf16():
    """SD-2.x prompt encoder: OpenCLIP ViT-H/14 text tower (1024 wide, 16 heads of 64, erf GELU), reduced depth for the CPU oracle;
    its 1024-wide output is the context width of configs.SD21_BASE."""
    from diffusion_pullback_amd import configs as cf
    from diffusion_pullback_amd.text_encoder import ClipTextEncoder
    from oracle import clip_text as oc
    assert cf.SD21_CLIP.hidden == cf.SD21_BASE.cross_dim == 1024 and cf.SD21_CLIP.layers == 23 and cf.SD21_CLIP.act == "gelu"
    assert cf.clip_config_for("stabilityai/stable-diffusion-2-1-base") is cf.SD21_CLIP and cf.clip_config_for("runwayml/stable-diffusion-v1-5") is cf.SD15_CLIP
    cfg = cf.CLIPTextConfig(vocab_size=1000, hidden=1024, layers=2, heads=16, intermediate=4096, max_position=77, act="gelu")
    p = cf.clip_init_params(cfg, seed=6)
    ids = torch.randint(0, cfg.vocab_size, (1, 77), generator=torch.Generator().manual_seed(1))
    for dtype, tol in ((torch.bfloat16, 3e-2), (torch.float16, 5e-3)):
        enc = ClipTextEncoder(cfg, p, dtype=dtype, device="cuda:0", max_batch=1)
        ref = oc.encode({k: v.to(dtype).float() for k, v in p.items()}, cfg, ids.long())
        y = enc(ids).cpu()
        assert y.shape == (1, 77, 1024) and rel(y, ref) < tol, (dtype, rel(y, ref))
