"""Abstract plugin surface — same methods as the reference's src/index/search_index.py:1-24."""


class SearchIndex:
    def __init__(self, media_type, asset_id, assets):
        raise NotImplementedError

    def get_index_filename(self, index_type):
        raise NotImplementedError

    def create_index(self, index_type, overwrite=False):
        raise NotImplementedError

    def is_index_loaded(self):
        raise NotImplementedError

    def load_index(self, index_type):
        raise NotImplementedError

    def search(self, media_type, query, topk=5, query_type='text'):
        raise NotImplementedError
