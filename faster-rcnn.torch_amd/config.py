"""Hyper-parameter tables of the reference (config/duplo.lua:1-17, config/imagenet.lua:1-17) as
plain dicts; dataset paths are left empty (the data loader is out of scope)."""

duplo_cfg = dict(
    class_count=16, target_smaller_side=450, scales=[32, 64, 128, 256], max_pixel_size=1000,
    normalization=dict(method="contrastive", width=7, centering=True, scaling=True),
    augmentation=dict(vflip=0.5, hflip=0.5, random_scaling=0.0, aspect_jitter=0.0),
    color_space="yuv", roi_pooling=dict(kw=6, kh=6), examples_base_path="", background_base_path="",
    batch_size=256, positive_threshold=0.5, negative_threshold=0.25, best_match=True, nearby_aversion=True)

imgnet_cfg = dict(
    class_count=200, target_smaller_side=480, scales=[48, 96, 192, 384], max_pixel_size=1000,
    normalization=dict(method="contrastive", width=7, centering=True, scaling=True),
    augmentation=dict(vflip=0, hflip=0.25, random_scaling=0, aspect_jitter=0),
    color_space="yuv", roi_pooling=dict(kw=6, kh=6), examples_base_path="", background_base_path="",
    batch_size=300, positive_threshold=0.6, negative_threshold=0.25, best_match=True, nearby_aversion=True)
