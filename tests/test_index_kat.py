"""Hand-derived known-answer tests of FPS / ball query (tests/kat_index.py): the C oracle here (CPU), the HIP kernels on
the GPU box -- both against vectors worked out from the reference .cu text, not from each other."""
import numpy as np
import pytest
import torch

from kat_index import BALLQ, BALLQ_F64, FPS

T = torch.from_numpy


@pytest.mark.parametrize("case", FPS, ids=[c["name"] for c in FPS])
def test_fps_kat_oracle(case):
    from oracle import index_ref
    idx = index_ref.furthest_point_sampling(T(case["xyz"]), case["m"])
    assert idx.tolist() == [case["idx"]]


@pytest.mark.parametrize("case", BALLQ, ids=[c["name"] for c in BALLQ])
def test_ball_query_kat_oracle(case):
    from oracle import index_ref
    idx = index_ref.ball_query(T(case["query"]), T(case["support"]), case["radius"], case["nsample"])
    assert idx.tolist() == [case["idx"]]


@pytest.mark.gpu
@pytest.mark.parametrize("case", FPS, ids=[c["name"] for c in FPS])
def test_fps_kat_hip(gpu, vgtk_alias, case):
    import vgtk.cuda.grouping as cuda_nn
    idx = cuda_nn.furthest_point_sampling(T(case["xyz"]).to(gpu), case["m"])
    assert idx.cpu().tolist() == [case["idx"]]


@pytest.mark.gpu
@pytest.mark.parametrize("case", BALLQ, ids=[c["name"] for c in BALLQ])
def test_ball_query_kat_hip(gpu, vgtk_alias, case):
    import vgtk.cuda.grouping as cuda_nn
    idx = cuda_nn.ball_query(T(case["query"]).to(gpu), T(case["support"]).to(gpu), case["radius"], case["nsample"])
    assert idx.cpu().tolist() == [case["idx"]]


# ---- fp64 dispatch (AT_DISPATCH_FLOATING_TYPES in the reference): the same hand-derived vectors hold in double (their
#      coordinates are exact binary fractions and no decision sits on a float rounding boundary)
@pytest.mark.parametrize("case", FPS, ids=[c["name"] for c in FPS])
def test_fps_kat_oracle_f64(case):
    from oracle import index_ref
    idx = index_ref.furthest_point_sampling(T(case["xyz"]).double(), case["m"])
    assert idx.tolist() == [case["idx"]]


@pytest.mark.parametrize("case", BALLQ + BALLQ_F64, ids=[c["name"] for c in BALLQ + BALLQ_F64])
def test_ball_query_kat_oracle_f64(case):
    from oracle import index_ref
    idx = index_ref.ball_query(T(case["query"]).double(), T(case["support"]).double(), case["radius"], case["nsample"])
    assert idx.tolist() == [case["idx"]]


@pytest.mark.gpu
@pytest.mark.parametrize("case", FPS, ids=[c["name"] for c in FPS])
def test_fps_kat_hip_f64(gpu, vgtk_alias, case):
    import vgtk.cuda.grouping as cuda_nn
    idx = cuda_nn.furthest_point_sampling(T(case["xyz"]).double().to(gpu), case["m"])
    assert idx.cpu().tolist() == [case["idx"]]


@pytest.mark.gpu
@pytest.mark.parametrize("case", BALLQ + BALLQ_F64, ids=[c["name"] for c in BALLQ + BALLQ_F64])
def test_ball_query_kat_hip_f64(gpu, vgtk_alias, case):
    import vgtk.cuda.grouping as cuda_nn
    idx = cuda_nn.ball_query(T(case["query"]).double().to(gpu), T(case["support"]).double().to(gpu), case["radius"],
                             case["nsample"])
    assert idx.cpu().tolist() == [case["idx"]]
