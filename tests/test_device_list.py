"""CPU: host/device_list.h — the device list the in-tree Maps read from RX_GPU_VECTOR_INDEXES (SURVEY §8e "Host topology": the reference's
`Map(metric, dim, maxElements)` constructor, hnsw_index.cc:61-66, has no device argument)."""
import pytest

from reindexer_amd import hostapi


@pytest.mark.parametrize("text,want", [
    ("3", [3]), ("0,1,2,3", [0, 1, 2, 3]), ("0-7", list(range(8))), ("0-3,6,7", [0, 1, 2, 3, 6, 7]), (" 0 , 1 ", [0, 1]),
    ("0,0,0", [0, 0, 0]), ("2-2", [2]),
    (None, []), ("", []), ("  ", []), ("gpu", []), ("0,", []), (",0", []), ("3-1", []), ("-1", []), ("0-", []), ("1;2", []), ("0,x", []),
    ("99999", []),
])
def test_parse_device_list(text, want):
    assert hostapi.parse_device_list(text) == want


def test_more_devices_than_shard_slots_is_refused():
    assert hostapi.parse_device_list("0-63") == list(range(64))
    assert hostapi.parse_device_list("0-64") == []
