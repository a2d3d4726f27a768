"""The implicit coordinate manager of `ME.SparseTensor(feats, ...)` with no `coordinate_manager` argument -- the idiom of
/root/reference/models/convolutional/lossy_coord/model.py (SHARE_COORDINATE_MANAGER + tensors built without a manager) -- in both
operation modes, and the per-thread scope of the "global" manager (fastpcc_amd/serving.py keeps one frame per thread)."""
import threading

import pytest
import torch

from fastpcc_amd import engine as ME


@pytest.fixture(autouse=True)
def _restore_mode():
    before = ME._operation_mode
    ME.clear_global_coordinate_manager()
    yield
    ME.set_sparse_tensor_operation_mode(before)
    ME.clear_global_coordinate_manager()


def _manager_with_a_map(rows: int):
    cm = ME.CoordinateManager()
    key = cm._register(ME._Map(0, 10, rows, None), 'm')
    return cm, key


def test_shared_mode_uses_the_threads_global_manager():
    ME.set_sparse_tensor_operation_mode(ME.SparseTensorOperationMode.SHARE_COORDINATE_MANAGER)
    cm, key = _manager_with_a_map(5)
    ME.set_global_coordinate_manager(cm)
    t = ME.SparseTensor(torch.zeros(5, 3), coordinate_map_key=key)           # no manager given
    assert t.coordinate_manager is cm


def test_shared_mode_creates_and_publishes_a_manager_when_none_exists():
    ME.set_sparse_tensor_operation_mode(ME.SparseTensorOperationMode.SHARE_COORDINATE_MANAGER)
    assert ME.global_coordinate_manager() is None
    with pytest.raises(KeyError):                                            # the fresh manager does not hold the key ...
        ME.SparseTensor(torch.zeros(5, 3), coordinate_map_key=ME.CoordinateMapKey(1, 'm'))
    assert isinstance(ME.global_coordinate_manager(), ME.CoordinateManager)  # ... but it was created and published


def test_separate_mode_never_touches_the_global_manager():
    ME.set_sparse_tensor_operation_mode(ME.SparseTensorOperationMode.SEPARATE_COORDINATE_MANAGER)
    cm, key = _manager_with_a_map(5)
    ME.set_global_coordinate_manager(cm)
    with pytest.raises(KeyError):                                            # a manager of its own, which has no such map
        ME.SparseTensor(torch.zeros(5, 3), coordinate_map_key=key)
    assert ME.global_coordinate_manager() is cm


def test_a_worker_thread_has_its_own_global_manager():
    ME.set_sparse_tensor_operation_mode(ME.SparseTensorOperationMode.SHARE_COORDINATE_MANAGER)
    cm, key = _manager_with_a_map(4)
    ME.set_global_coordinate_manager(cm)
    seen = {}

    def worker():
        seen['before'] = ME.global_coordinate_manager()
        mine, k2 = _manager_with_a_map(7)
        ME.set_global_coordinate_manager(mine)
        seen['tensor_cm'] = ME.SparseTensor(torch.zeros(7, 1), coordinate_map_key=k2).coordinate_manager
        seen['mine'] = mine

    th = threading.Thread(target=worker)
    th.start()
    th.join()
    assert seen['before'] is None and seen['tensor_cm'] is seen['mine'] and seen['mine'] is not cm
    assert ME.global_coordinate_manager() is cm                              # the worker did not disturb this thread's manager


@pytest.mark.gpu
@pytest.mark.parametrize('mode', [ME.SparseTensorOperationMode.SHARE_COORDINATE_MANAGER,
                                  ME.SparseTensorOperationMode.SEPARATE_COORDINATE_MANAGER])
def test_tensor_from_coordinates_without_a_manager(mode):
    ME.set_sparse_tensor_operation_mode(mode)
    coords = torch.tensor([[0, 3, 1, 2], [0, 0, 0, 0], [0, 1, 1, 1]], dtype=torch.int32, device='cuda')
    a = ME.SparseTensor(torch.arange(3, dtype=torch.float32, device='cuda').view(3, 1), coordinates=coords)
    b = ME.SparseTensor(torch.ones(3, 1, device='cuda'), coordinates=coords)
    shared = mode == ME.SparseTensorOperationMode.SHARE_COORDINATE_MANAGER
    assert (a.coordinate_manager is b.coordinate_manager) == shared
    assert (ME.global_coordinate_manager() is a.coordinate_manager) == shared
    assert a.C.shape == (3, 4) and a.F.view(-1).tolist() == [1.0, 2.0, 0.0]   # Morton order: (0,0,0), (1,1,1), (3,1,2)


# ---- batches of independent clouds: the host-side bookkeeping (no device work) ---------------------------------------------------------
def test_cloud_ranges_of_generated_and_refined_maps_need_no_device():
    cm = ME.CoordinateManager(clouds=3)
    assert cm.independent_clouds and cm._n_batch == 3
    top = ME._Map(3, 7, 10, None)
    top.edges = [0, 4, 4, 10]                                               # cloud 1 has no rows on this level
    cm._register(top, 'top')
    gen = ME._Map(2, 8, 80, None)
    gen.parent, gen.generated = top, True
    assert cm.batch_offsets(gen) == [0, 32, 32, 80] and cm.cloud_rows(gen) == [32, 0, 48]
    assert cm.cloud_rows(top) == [4, 0, 6]
    single = ME.CoordinateManager()
    single._n_batch = 1
    m = ME._Map(0, 10, 9, None)
    assert single.batch_offsets(m) == [0, 9] and single.cloud_rows(m) == [9] and not single.independent_clouds
    assert ME._edges_of([3, 0, 2]) == [0, 3, 3, 5]
    with pytest.raises(ValueError):
        ME.CoordinateManager(clouds=0)


def test_pad_rule_is_asked_per_cloud():
    """PAD_MIN_ROWS decides the summation order of the narrow layers; in a batch of independent clouds it looks at each cloud's rows"""
    assert ME._pad_plan(16, 0, 8, ME.PAD_MIN_ROWS) == (16, 0, 32) and ME._pad_plan(16, 0, 8, ME.PAD_MIN_ROWS - 1) is None
    cm = ME.CoordinateManager(clouds=2)
    m = ME._Map(0, 10, 2 * ME.PAD_MIN_ROWS - 2, None)
    m.edges = [0, ME.PAD_MIN_ROWS - 1, 2 * ME.PAD_MIN_ROWS - 2]
    # the union has more than PAD_MIN_ROWS rows, neither cloud has: both clouds are "small"
    assert cm.cloud_rows(m) == [ME.PAD_MIN_ROWS - 1] * 2 and m.n >= ME.PAD_MIN_ROWS


def test_partition_groups():
    from fastpcc_amd.codecs.lossy_coord_v2.model import PCC
    g = PCC._groups
    class Fake:
        MANY_MAX_VOXELS = 100
    assert g(Fake, [60, 30, 20, 100, 1, 1]) == [[0, 1], [2], [3], [4, 5]]
    assert g(Fake, [500]) == [[0]] and g(Fake, []) == []
    Fake.MANY_MAX_VOXELS = 10 ** 9
    assert g(Fake, list(range(70))) == [list(range(64)), list(range(64, 70))]       # the engine's limit of 64 clouds per batch
