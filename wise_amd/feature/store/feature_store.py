"""Abstract FeatureStore — same surface as the reference's src/feature/store/feature_store.py:1-14."""


class FeatureStore:
    def __init__(self, store_name, store_data_dir):
        raise NotImplementedError

    def add(self, index, features):
        raise NotImplementedError

    def load(self, start_index=0, count=-1):
        raise NotImplementedError
