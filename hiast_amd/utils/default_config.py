"""Configuration tree with the reference's keys and defaults (utils/default_config.py:3-182) on a small
attribute-tree node that restates the yacs.CfgNode behaviour the reference relies on:
merge_from_file / merge_from_list (unknown keys are errors), string values decoded with
ast.literal_eval (so YAML `3e-6` and `None` arrive as float / None), freeze(), dump(), clone().
yacs itself is not available offline."""
import ast
import copy

import yaml


class CfgNode(dict):
    _FROZEN = "__frozen__"

    def __init__(self, init=None):
        super().__init__()
        object.__setattr__(self, CfgNode._FROZEN, False)
        for k, v in (init or {}).items():
            dict.__setitem__(self, k, CfgNode(v) if isinstance(v, dict) else v)

    # attribute access
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        if object.__getattribute__(self, CfgNode._FROZEN):
            raise AttributeError("attempted to set %s on a frozen CfgNode" % name)
        self[name] = CfgNode(value) if isinstance(value, dict) and not isinstance(value, CfgNode) else value

    def is_frozen(self):
        return object.__getattribute__(self, CfgNode._FROZEN)

    def _set_frozen(self, flag):
        object.__setattr__(self, CfgNode._FROZEN, flag)
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_frozen(flag)

    def freeze(self):
        self._set_frozen(True)

    def defrost(self):
        self._set_frozen(False)

    def clone(self):
        c = CfgNode(self.to_dict())
        return c

    def __deepcopy__(self, memo):
        return self.clone()

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, CfgNode) else copy.deepcopy(v)) for k, v in self.items()}

    def dump(self):
        return yaml.safe_dump(self.to_dict(), default_flow_style=None, sort_keys=True)

    def __str__(self):
        return self.dump()

    @staticmethod
    def _decode(v):
        if isinstance(v, str):
            try:
                return ast.literal_eval(v)
            except (ValueError, SyntaxError):
                return v
        return v

    def _merge(self, other, path):
        if self.is_frozen():
            raise AttributeError("cannot merge into a frozen CfgNode")
        for k, v in other.items():
            full = ".".join(path + [k])
            if k not in self:
                raise KeyError("Non-existent config key: %s" % full)
            if isinstance(self[k], CfgNode):
                if not isinstance(v, dict):
                    raise ValueError("%s must be a mapping" % full)
                self[k]._merge(v, path + [k])
            else:
                v = self._decode(v)
                old = self[k]
                if old is not None and v is not None and type(old) is not type(v):
                    if isinstance(old, float) and isinstance(v, int) and not isinstance(v, bool):
                        v = float(v)
                    elif isinstance(old, (list, tuple)) and isinstance(v, (list, tuple)):
                        v = type(old)(v)
                    else:
                        raise ValueError("Type mismatch (%s vs. %s) for config key: %s" % (type(old), type(v), full))
                dict.__setitem__(self, k, v)

    def merge_from_file(self, path):
        with open(path) as f:
            self._merge(yaml.safe_load(f) or {}, [])

    def merge_from_other_cfg(self, other):
        self._merge(other.to_dict() if isinstance(other, CfgNode) else other, [])

    def merge_from_list(self, kv):
        assert len(kv) % 2 == 0
        for key, v in zip(kv[0::2], kv[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                node = node[p]
            node._merge({parts[-1]: v}, parts[:-1])


_DEFAULTS = yaml.safe_load("""
trainer: null
work_dir: './'
model:
  type: null
  is_freeze_bn: true              # should be True after source-only training
  seg_model: {type: DeepLab_V2, output_dim: 256}
  predictor:
    seg_loss: {type: CE, source_weight: 1.0, target_pseudo_weight: 1.0}
    kld_loss: {weight: 0.1}       # confident region, self-training
    ent_loss: {weight: 3.0}       # ignored region, self-training
  discriminator:
    is_enabled: false
    is_entropy_input: false
    lr: 0.0001
    D_loss: {type: MSE, weight: 1.0, adv_weight: 0.05}
dataset:
  num_classes: 19
  num_workers: 2
  decoded_cache_dir: null         # (not in the reference) directory for decoded uint8 arrays, e.g. /dev/shm/hiast_cache
  decoded_cache_gb: 16.0
  source: {type: null, json_path: null, image_dir: null, aug_type: []}
  target: {type: null, json_path: null, image_dir: null, pseudo_dir: null, aug_type: []}
  val: {type: null, json_path: null, image_dir: null, resize_size: null}
pseudo_policy:
  resume_from: null
  batch_size: 2
  resize_size: null               # [height, width]
  save_dir: null
  type: null                      # IAS, CBST, CT, NT
  ias: {alpha: 0.2, beta: 0.9, gamma: 8.0}
  cbst: {p: 0.2, sample_interval: 4}
  ct: {threshold: 0.9}
train:
  batch_size: 4
  lr: 0.0001
  optimizer: Adam
  resume_from: null
  apex_opt: O1
  amp_dtype: fp16                 # (not in the reference) 16-bit type of the O1+ mixed-precision step, both on the same
                                  # hand-written channels-last kernels: fp16 = the reference's apex-O1 arithmetic (dynamic
                                  # loss scaling, decided on the device) | bf16 = 8 exponent bits, no loss scaling
  gpu_num: 2
  random_seed: 888
  port: 6789
  is_save_all: false
  is_debug: false
  total_iter: 10000
  iter_report: 100
  iter_val: 400
  lr_scheduler:
    type: Cosine
    poly: {power: 0.9}
validate:
  resume_from: null
  resize_sizes: []
  is_flip: false
  batch_size: 2
  color_mask_dir_path: null
cst_training:
  is_enabled: false
  ema_model: {iter_update: 1, gamma: 0.999}
  cst_loss: {type: SoftCE, weight: 1.0, region: ignored}
mut_training:
  is_enabled: false
  resume_from: null
  is_strong_input: false
  mut_loss: {weight: 0.1, region: ignored}
preprocessor:
  type: null
  copy_paste: {mode: original, name: normal, selected_num_classes: 14, gamma: 0.99}
""")


def get_default_cfg():
    """a fresh, unfrozen copy of the default tree"""
    return CfgNode(_DEFAULTS)


cfg = get_default_cfg()
