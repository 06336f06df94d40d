"""SearchIndexFactory — same dispatch and errors as the reference's
src/index/search_index_factory.py:4-21.  The metadata (SQLite FTS) index is outside this build."""
from .feature_search_index import FeatureSearchIndex


def SearchIndexFactory(media_type, asset_id, asset):
    if media_type in ['audio', 'video', 'image']:
        return FeatureSearchIndex(media_type, asset_id, asset)
    elif media_type == 'metadata':
        raise NotImplementedError('SqliteSearchIndex (full-text metadata search) is outside the MI355X hot path; '
                                  'use the reference implementation for media_type="metadata"')
    else:
        raise ValueError(f'Unknown media_type {media_type}')
