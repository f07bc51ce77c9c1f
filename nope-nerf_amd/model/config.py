"""get_model(renderer, cfg, device) -> nn.Module, signature of reference model/config.py:4-18.
The optional DPT mono-depth estimator is an offline preprocessing network (every training config sets depth.type
None); requesting it raises, because the vendored DPT stack is outside the hot path."""
import model as mdl


def get_model(renderer, cfg, device=None, **kwargs):
    if cfg['depth']['type'] == 'DPT':
        raise NotImplementedError("depth.type == 'DPT' builds the vendored DPT network (offline preprocessing, reference "
                                  "preprocess/dpt_depth.py); run that step with the reference and train with depth.type None")
    return mdl.nope_nerf(cfg, renderer, None, device)
