"""CPU: pin the oracle (oracle/conformer.py) against fixtures produced by the reference itself
(tests/golden/make_golden.py).  Tolerances are fp32 round-off of a different summation order."""
import numpy as np
import pytest
import torch

from oracle import conformer as oc
from oracle.mel import slaney_mel_filterbank
from tests.helpers import golden_speech, golden_state_dict, hparams, load_golden

torch.set_grad_enabled(False)


def test_melmat_matches_reference_buffer():
    g = load_golden("small_10s")
    m = slaney_mel_filterbank(sr=16000, n_fft=512, n_mels=80, fmin=0, fmax=8000).T
    assert m.shape == (257, 80)
    np.testing.assert_array_equal(m, g["melmat"])
    # every rFFT bin feeds at most two neighbouring filters (triangles overlap by one)
    assert ((m > 0).sum(1) <= 2).all()


def test_stft_olens_formula():
    # test/espnet2/layers/test_stft.py:10-16: n_fft 4 / hop 2, ilens [30,15] -> olens [16,8]
    sp = torch.randn(2, 30)
    lens = torch.tensor([30, 15])
    mel = torch.rand(3, 2)
    feats, olens = oc.frontend_feats(sp, lens, mel, n_fft=4, win_length=4, hop=2)
    assert olens.tolist() == [16, 8]
    assert feats.shape == (2, 16, 2)
    assert (feats[1, 8:] == 0).all()


@pytest.mark.parametrize("name", ["tiny_blocks", "small_ragged", "small_10s", "large_10s"])
def test_frontend_and_encoder_match_reference(name):
    g = load_golden(name)
    sd = golden_state_dict(g)
    hp = hparams(g)
    speech, lens = golden_speech(g)
    feats, flens = oc.frontend_feats(speech, lens, sd["frontend.logmel.melmat"], hp["n_fft"],
                                     hp["win_length"], hp["hop"])
    assert flens.tolist() == g["feats_lens"].tolist()
    np.testing.assert_allclose(feats.numpy(), g["feats"], atol=2e-4, rtol=0)
    enc, olens = oc.encode(sd, speech, lens, hp["heads"], hp["num_blocks"], hp["n_fft"],
                           hp["win_length"], hp["hop"])
    assert olens.tolist() == g["enc_olens"].tolist()
    ke = int(g["enc_keep_every"])
    np.testing.assert_allclose(enc[:, ::ke].numpy(), g["enc_out"], atol=5e-4, rtol=0)
    ids = oc.ctc_argmax(sd, enc).numpy()
    # argmax can only differ where the reference's own top-2 margin is within round-off
    diff = ids != g["ctc_ids"]
    assert (g["ctc_margin"][diff] < 1e-4).all()
    assert diff.mean() < 0.01
    if not diff.any():
        eos = int(g["vocab"]) - 1
        toks = oc.greedy_ctc(sd, enc, olens, blank=0, sos_eos=eos)
        for b, t in enumerate(toks):
            assert t == g["g1_tokens"][b, : g["g1_lens"][b]].tolist()


def test_block_outputs_match_reference():
    g = load_golden("tiny_blocks")
    sd = golden_state_dict(g)
    hp = hparams(g)
    speech, lens = golden_speech(g)
    feats, flens = oc.frontend_feats(speech, lens, sd["frontend.logmel.melmat"], hp["n_fft"],
                                     hp["win_length"], hp["hop"])
    feats = oc.utterance_mvn(feats, flens)
    x = oc.conv2d_subsampling(sd, feats) * (64 ** 0.5)
    np.testing.assert_allclose(x.numpy(), g["embed_x"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(oc.rel_pos_emb(x.size(1), 64).numpy(), g["pos_emb"][0], atol=1e-6)
    _, _, blocks = oc.conformer_encoder(sd, feats, flens, hp["heads"], hp["num_blocks"],
                                        return_blocks=True)
    for i, b in enumerate(blocks):
        np.testing.assert_allclose(b.numpy(), g["block_outs"][i], atol=1e-4, rtol=0)


def test_rel_shift_identity():
    # rel_shift(BD)[i, j] = BD[i, T-1-i+j]
    T = 9
    bd = torch.randn(1, 2, T, 2 * T - 1)
    out = oc.rel_shift(bd)
    i = torch.arange(T)[:, None]
    j = torch.arange(T)[None, :]
    ref = bd[:, :, i, T - 1 - i + j]
    assert torch.equal(out, ref)


def test_too_short():
    g = load_golden("tiny_blocks")
    sd = golden_state_dict(g)
    with pytest.raises(oc.TooShortUttError):
        oc.conformer_encoder(sd, torch.zeros(1, 6, 80), torch.tensor([6]), 1, 2)
