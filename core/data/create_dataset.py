"""create_dataloader(data_type): the reference's data entry point (core/data/create_dataset.py:
59-74).  The ZJU-MoCap / OcMotion pickles and the SMPL model are not redistributable, so this
build ships only the synthetic subject (occnerf_amd/synth.py): same per-frame dict, same shapes,
yielded with the leading batch dimension a torch DataLoader with batch_size=1 would add
(run.py strips it, run.py:85-86)."""
from configs import cfg
from occnerf_amd.sequence import SyntheticFrames


def create_dataloader(data_type='train', evaluate=False, **_):
    if cfg.get('dataset', 'synthetic') != 'synthetic' or \
            data_type not in ('tpose', 'freeview', 'movement', 'allview', 'progress'):
        raise NotImplementedError(
            f"dataset '{cfg.get('dataset')}' / type '{data_type}': only the synthetic tpose / freeview / movement / "
            'allview / progress frame generators ship with this build (datasets are out of scope, SURVEY.md 2)')
    return SyntheticFrames(data_type, img_size=int(cfg.get('render_size', 512)), render_frames=int(cfg.render_frames),
                           bgcolor=cfg.bgcolor, device_rays=bool(cfg.get('device_rays', True)),
                           freeview_frame_idx=int(cfg.freeview.get('frame_idx', 0)))
