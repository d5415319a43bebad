"""GeoTIFF container parsing and geo tags (no GPU): topo_geotiff_info against fixtures built by tests/tiff_writer.py and by
Pillow/libtiff; the error cases of CoordinateTransform::from_geo_tag_data and of malformed / unsupported files."""
import io

import numpy as np
import pytest

from tiff_writer import write_geotiff


def _arr(w=37, h=23, seed=1):
    rng = np.random.default_rng(seed)
    return (rng.normal(1500.0, 700.0, (h, w))).astype(np.float32)


@pytest.mark.parametrize("kw", [
    dict(), dict(byteorder=">"), dict(compression="none", predictor=1), dict(tile=(16, 16)), dict(rows_per_strip=5, predictor=2),
    dict(compression="packbits", predictor=1, rows_per_strip=1),
])
def test_info_of_written_files(topo, kw):
    a = _arr()
    w, h, ct = topo.geotiff_info(write_geotiff(a, tie_points=(0.5, 0.25, 0, 10.0 - 1 / 2400, 48.0 + 1 / 2400, 0), **kw))
    assert (w, h) == (a.shape[1], a.shape[0])
    assert ct.raster_point.tolist() == [0.5, 0.25]
    assert ct.model_point.tolist() == [float(np.float32(10.0 - 1 / 2400)), float(np.float32(48.0 + 1 / 2400))]
    assert ct.pixel_scale.tolist() == [float(np.float32(1 / 1200))] * 2


def test_geo_tag_errors_follow_the_reference(topo):
    a = _arr()
    with pytest.raises(topo.TopoError) as e:      # ModelTransformationTag present -> IncorrectGeoTags
        topo.geotiff_info(write_geotiff(a, model_transformation=[1.0] * 16))
    assert e.value.code == -2
    for kw in (dict(pixel_scale=None), dict(tie_points=None)):
        with pytest.raises(topo.TopoError) as e:
            topo.geotiff_info(write_geotiff(a, **kw))
        assert e.value.code == -2
    for kw in (dict(pixel_scale=(1.0, 1.0)), dict(tie_points=(0, 0, 0, 1.0, 2.0))):      # IncorrectGeoTagData
        with pytest.raises(topo.TopoError) as e:
            topo.geotiff_info(write_geotiff(a, **kw))
        assert e.value.code == -1


def test_malformed_and_unsupported_files(topo):
    good = write_geotiff(_arr())
    for bad in (b"", b"II*\0", b"XX*\0" + good[4:], good[:40], good[:2] + b"\x2b\0" + good[4:]):
        with pytest.raises(topo.TopoError):
            topo.geotiff_info(bad)
    with pytest.raises(topo.TopoError) as e:      # BigTIFF
        topo.geotiff_info(good[:2] + b"\x2b\0" + good[4:])
    assert e.value.code == -2
    for kw in (dict(bits=16), dict(sample_format=1)):      # not DecodingResult::F32
        with pytest.raises(topo.TopoError) as e:
            topo.geotiff_info(write_geotiff(_arr(), **kw))
        assert e.value.code == -2
    # a strip table that points past the end of the file
    cut = good[:len(good) // 2]
    with pytest.raises(topo.TopoError):
        topo.geotiff_info(cut)


def test_info_of_libtiff_files(topo):
    """Files written by an independent encoder (Pillow -> libtiff), with the geo tags attached as TIFF tag data."""
    PIL = pytest.importorskip("PIL")
    from PIL import Image, TiffImagePlugin
    a = _arr(64, 48)
    for comp in (None, "tiff_adobe_deflate", "tiff_lzw", "packbits"):
        ifd = TiffImagePlugin.ImageFileDirectory_v2()
        ifd[33550] = (1 / 1200, 1 / 1200, 0.0)
        ifd.tagtype[33550] = 12
        ifd[33922] = (0.0, 0.0, 0.0, 11.0, 47.0, 0.0)
        ifd.tagtype[33922] = 12
        buf = io.BytesIO()
        Image.fromarray(a, mode="F").save(buf, format="TIFF", compression=comp, tiffinfo=ifd)
        w, h, ct = topo.geotiff_info(buf.getvalue())
        assert (w, h) == (64, 48) and ct.model_point.tolist() == [11.0, 47.0]
