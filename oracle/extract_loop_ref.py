"""ORACLE (test infrastructure — never imported by the product path under wise_amd/).

Sequential restatement of the reference's HP-1 driver loop, /root/reference/extract-features.py:324-375:
one extractor call per chunk, one `vectors` row per embedding (ids from an autoincrement counter that
video and audio share, starting at 1), video/image timestamps pts + i*0.5 (:353), audio [pts, pts+4] (:362-363),
audio chunks shorter than 192000 samples dropped (:336-338), store.add per vector in creation order.
"""
import numpy as np


def reference_loop(loader, feature_extractors, feature_stores, create_vector, video_frame_rate=2,
                   audio_segment_length=4.0, audio_frames_per_chunk=192000):
    for idx, (mid, chunks) in enumerate(loader):
        for media_type in feature_extractors:
            if media_type not in chunks or chunks[media_type] is None:
                continue
            segment_tensor = chunks[media_type].tensor
            segment_pts = chunks[media_type].pts
            if media_type == "image" or media_type == "video":
                segment_feature = feature_extractors[media_type].extract_image_features(segment_tensor)
            elif media_type == "audio":
                if segment_tensor.shape[2] < audio_frames_per_chunk:
                    continue
                segment_feature = feature_extractors[media_type].extract_audio_features(segment_tensor)
            else:
                raise ValueError("Unknown media_type {media_type}")
            if media_type in ("video", "image"):
                for i in range(len(segment_feature)):
                    vid = create_vector(media_type, mid, segment_pts + i * (1 / video_frame_rate), None)
                    feature_stores[media_type].add(vid, np.expand_dims(segment_feature[i], axis=0))
            else:
                vid = create_vector(media_type, mid, segment_pts, segment_pts + audio_segment_length)
                feature_stores[media_type].add(vid, segment_feature)
    for store in feature_stores.values():
        store.close()
