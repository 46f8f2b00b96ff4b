"""CPU checks of the encoder mirrors: state-dict contract vs the reference's module tree (names,
order and shapes recorded in tests/golden/encoder_golden.npz), weight packing, error paths."""
import numpy as np
import pytest
import torch

from zeroshape_amd.utils.options import EasyDict as edict


def make_opt(encoder="resnet"):
    return edict(dict(H=224, W=224, device="cpu", pretrain=dict(depth=None),
                      arch=dict(num_heads=8, latent_dim=256, win_size=16,
                                depth=dict(encoder=encoder, n_blocks=12, dsp=2, pretrained=None),
                                rgb=dict(encoder=None, n_blocks=12),
                                impl=dict(n_channels=256, att_blocks=2, mlp_ratio=4., posenc_perlayer=False,
                                          mlp_layers=8, posenc_3D=0, skip_in=[2, 4, 6]))))


def contract(keys, shapes):
    return [(str(k), tuple(int(x) for x in str(s).split(",") if x)) for k, s in zip(keys, shapes)]


def test_graph_state_dict_equals_reference(encoder_golden, encoder_sd):
    from zeroshape_amd.model.compute_graph.graph_shape import Graph
    opt = make_opt()
    g = Graph(opt)
    assert opt.arch.depth.dsp == 1                       # graph_shape.py:42 side effect
    mine = [(k, tuple(v.shape)) for k, v in g.state_dict().items()]
    assert mine == contract(encoder_golden["graph_keys"], encoder_golden["graph_shapes"])
    # a reference checkpoint's sub-dicts load by name, strictly (graph_shape.py:75-79)
    from zeroshape_amd.utils.util import get_child_state_dict
    ckpt = {"module.graph." + k if i % 2 else "graph." + k: v for i, (k, v) in enumerate(encoder_sd.items())}
    ckpt = get_child_state_dict(ckpt, "graph")
    g.dpt_depth.load_state_dict(get_child_state_dict(ckpt, "dpt_depth"), strict=True)
    g.intr_head.load_state_dict(get_child_state_dict(ckpt, "intr_head"), strict=True)
    g.intr_proj.load_state_dict(get_child_state_dict(ckpt, "intr_proj"), strict=True)
    g.coord_encoder.load_state_dict(get_child_state_dict(ckpt, "coord_encoder"), strict=True)
    assert bool((g.intr_proj.weight == encoder_sd["intr_proj.weight"]).all())


def test_intr_proj_starts_at_zero_and_head_bias():
    from zeroshape_amd.model.compute_graph.graph_shape import Graph
    g = Graph(make_opt())
    assert float(g.intr_proj.weight.abs().sum()) == 0 and float(g.intr_proj.bias.abs().sum()) == 0   # :26-28
    assert float(g.dpt_depth.scratch.output_conv[4].bias) == pytest.approx(0.05)                    # dpt_depth.py:108


def test_coord_enc_att_contract(encoder_golden):
    from zeroshape_amd.model.shape.seen_coord_enc import CoordEncAtt
    a = CoordEncAtt(embed_dim=256, n_blocks=12, num_heads=8, win_size=8)
    mine = [(k, tuple(v.shape)) for k, v in a.state_dict().items()]
    assert mine == contract(encoder_golden["att_keys"], encoder_golden["att_shapes"])
    # fixed window-local sin-cos embedding as the reference initialises it (:41-42)
    from zeroshape_amd.utils.pos_embed import get_2d_sincos_pos_embed
    pe = get_2d_sincos_pos_embed(256, 8, cls_token=True)
    assert np.array_equal(a.coord_embed.two_d_pos_embed[0].numpy(), pe.astype(np.float32))
    g_opt = make_opt("transformer")
    from zeroshape_amd.model.compute_graph.graph_shape import Graph
    g = Graph(g_opt)
    assert isinstance(g.coord_encoder, CoordEncAtt) and g.coord_encoder.win_size == 8 and g_opt.arch.depth.dsp == 2


def test_unsupported_configurations_raise():
    from zeroshape_amd.model.compute_graph.graph_shape import Graph
    from zeroshape_amd.model.depth.dpt_depth import DPTDepthModel
    with pytest.raises(NotImplementedError):
        DPTDepthModel(backbone="vitl16_384")
    opt = make_opt()
    opt.arch.rgb.encoder = "resnet"
    with pytest.raises(NotImplementedError):
        Graph(opt)
    opt = make_opt()
    opt.arch.win_size = 8
    with pytest.raises(NotImplementedError):
        Graph(opt)
    g = Graph(make_opt())
    var = edict(dict(idx=[0], rgb_input_map=torch.zeros(1, 3, 224, 224), mask_input_map=torch.zeros(1, 1, 224, 224)))
    with pytest.raises(ValueError):                       # CPU tensors: there is no CPU path, training or not
        g.forward(make_opt(), var, training=True)
    with pytest.raises(ValueError):
        g.forward(make_opt(), var, training=False, get_loss=False)
    from zeroshape_amd.utils.loss import Loss
    lopt = edict(training=dict(shape_loss=dict(impt_weight=1, impt_thres=0.01),
                               depth_loss=dict(grad_reg=0.1, depth_inv=True, mask_shrink=True)))
    with pytest.raises(ValueError):                       # eroded masks run on the GPU too: no CPU path
        Loss(lopt).depth_loss(torch.zeros(1, 1, 8, 8), torch.zeros(1, 1, 8, 8), torch.zeros(1, 1, 8, 8))
    with pytest.raises(NotImplementedError):              # train mode: BN folding is eval-only
        g.dpt_depth.train().packed("cpu")


def test_pack_weight_layout():
    """[K16/4][CoutPad][4] with k = (ky*kw + kx)*Cin + cin, zero padded (include/zeroshape_hip.h)."""
    from zeroshape_amd.nn import pack
    w = torch.arange(5 * 8 * 3 * 3, dtype=torch.float32).reshape(5, 8, 3, 3)
    flat, cin = pack.pack_weight(w)
    assert cin == 8 and flat.numel() == 80 * 128                      # K = 72 -> 80, Cout 5 -> 128
    q = flat.view(20, 128, 4)
    for (co, ci, ky, kx) in ((0, 0, 0, 0), (4, 7, 2, 2), (2, 5, 1, 0)):
        k = (ky * 3 + kx) * 8 + ci
        assert q[k // 4, co, k % 4] == w[co, ci, ky, kx]
    assert float(q[18:, :, :].abs().sum()) == 0 and float(q[:, 5:, :].abs().sum()) == 0
    flat, cin = pack.pack_weight(torch.ones(2, 3, 1, 1), cin_pad=4)
    assert cin == 4 and flat.view(4, 128, 4)[0, 0].tolist() == [1, 1, 1, 0]
    pc = pack.pack_conv(torch.ones(2, 4, 3, 3), stride=2, padding="same")
    assert pc.out_size(56, 3) == (28, 0) and pc.out_size(224, 3) == (112, 0)
    assert pack.pack_conv(torch.ones(2, 4, 7, 7), stride=2, padding="same").out_size(224, 7) == (112, 2)
    assert pack.pack_conv(torch.ones(2, 4, 3, 3), padding=1).out_size(14, 3) == (14, 1)
    scale, shift = pack.fold_bn(dict(weight=torch.tensor([2.0]), bias=torch.tensor([1.0]),
                                     running_mean=torch.tensor([3.0]), running_var=torch.tensor([4.0]), eps=0.0),
                                bias=torch.tensor([5.0]))
    assert float(scale) == 1.0 and float(shift) == 1.0 - 3.0 + 5.0
