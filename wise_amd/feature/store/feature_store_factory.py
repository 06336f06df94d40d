"""FeatureStoreFactory: pick a FeatureStore implementation by declared type or by what is on disk.

Mirrors the API of the reference's factory (src/feature/store/feature_store_factory.py:12-38):
`create_store(type, media_type, features_dir)` for the writer side (extract-features.py:64-71) and
`load_store(media_type, features_dir)` for the reader side, which infers the type from the shard files
`{media_type}-*.tar` / `{media_type}-*.npz` (feature_search_index.py:41).  Same exceptions: ValueError for an
unknown type, for a directory holding no shard of that media type, for mixed shard kinds and for a foreign extension.
"""
import enum
from pathlib import Path

from .numpy_save_store import NumpySaveStore
from .webdataset_store import WebdatasetStore


class FeatureStoreType(str, enum.Enum):
    WEBDATASET = "webdataset"
    NUMPY = "numpy"


# one table serves both directions: declared type -> class, shard extension -> class
_STORE_BY_TYPE = {FeatureStoreType.WEBDATASET: WebdatasetStore, FeatureStoreType.NUMPY: NumpySaveStore}
_STORE_BY_SUFFIX = {".tar": WebdatasetStore, ".npz": NumpySaveStore}


class FeatureStoreFactory:
    @classmethod
    def create_store(cls, feature_store_type: FeatureStoreType, media_type, features_dir):
        try:
            store_cls = _STORE_BY_TYPE[FeatureStoreType(feature_store_type)]
        except (ValueError, KeyError):
            raise ValueError(f'unknown feature_store_type {feature_store_type}') from None
        return store_cls(media_type, features_dir)

    @classmethod
    def load_store(cls, media_type, features_dir):
        features_dir = Path(features_dir)
        # non-recursive, like the reference's glob on `{media_type}-*.*`
        suffixes = sorted({p.suffix for p in features_dir.glob(f'{media_type}-*.*')})
        if len(suffixes) != 1:
            raise ValueError(f'failed to infer type of {media_type} feature store in {features_dir}')
        store_cls = _STORE_BY_SUFFIX.get(suffixes[0])
        if store_cls is None:
            raise ValueError(f'unknown store containing shard filenames with extension {suffixes[0]}')
        return store_cls(media_type, features_dir)
