"""FeatureExtractorFactory(id) — same dispatch and errors as the reference's
src/feature/feature_extractor_factory.py:4-27: the id must have exactly four '/' tokens, and the
prefix selects the implementation."""
from .microsoft_clap import MicrosoftClap
from .mlfoundation_openclip import MlfoundationOpenClip


def FeatureExtractorFactory(id):
    if len(id.split('/')) != 4:
        raise ValueError(f'''Feature extractor name must be formatted as
              MODEL_CREATOR_NAMESPACE / MODEL_CREATOR / MODEL_NAME / PRETRAINING_DATASET
            For example, use "mlfoundations/open_clip/ViT-B-32/openai" for extracting features using the
            OpenCLIP ViT-B/32 image tower.
            ''')
    if id.startswith('mlfoundations/open_clip/'):
        return MlfoundationOpenClip(id)
    elif id.startswith('microsoft/clap/'):
        return MicrosoftClap(id)
    else:
        raise ValueError(f'Unknown feature extractor id {id}')
