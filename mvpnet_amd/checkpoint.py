"""Checkpoint interop: reads and writes the reference's on-disk layout, so weights trained with
the reference evaluate here and vice versa.

Layout (common/utils/checkpoint.py:37-56,136-175):
  <save_dir>/<name>.pth      torch.save of {'model': state_dict, ['optimizer': ...], ['scheduler': ...], **extra}
                             (extra = e.g. iteration, best_metric_name value; train_mvpnet_3d.py saves model_{:06d} / model_best)
  <save_dir>/last_checkpoint text: `Checkpointer` one path, `CheckpointerV2` one path per line, oldest first; a path without
                             directory is relative to save_dir; V2 deletes the oldest file beyond max_to_keep.
The module wrappers the reference unwraps (DataParallel / DistributedDataParallel) are unwrapped here too; this build runs one
process per GPU, so a wrapper only appears when a user brings one.  Pinned by tests/golden/checkpoint_ref/ (files written by
the imported reference classes) and, in the build container, by the reference loading files written here (make_golden.py)."""
import hashlib
import logging
import os

import torch


def get_md5(filename):
    """common/utils/io.py:4-8"""
    digest = hashlib.md5()
    with open(filename, 'rb') as f:
        digest.update(f.read())
    return digest.hexdigest()


def _bare(model):
    return model.module if isinstance(model, (torch.nn.DataParallel, torch.nn.parallel.DistributedDataParallel)) else model


class Checkpointer(object):
    TAG = 'last_checkpoint'

    def __init__(self, model, optimizer=None, scheduler=None, save_dir='', logger=None):
        self.model = model
        self.optimizer = optimizer
        self.scheduler = scheduler
        self.save_dir = save_dir
        self.logger = logger
        self._print = logger.info if logger else print

    # ------------------------------------------------------------------ write
    def save(self, name, tag=True, **kwargs):
        if not self.save_dir:
            return
        data = {'model': _bare(self.model).state_dict()}
        for key, obj in (('optimizer', self.optimizer), ('scheduler', self.scheduler)):
            if obj is not None:
                data[key] = obj.state_dict()
        data.update(kwargs)
        save_file = os.path.join(self.save_dir, '{}.pth'.format(name))
        self._print('Saving checkpoint to {}'.format(os.path.abspath(save_file)))
        torch.save(data, save_file)
        if tag:
            self.tag_last_checkpoint(save_file)

    def tag_last_checkpoint(self, last_filename):
        with open(self._tag_file(), 'w') as f:
            f.write(last_filename if os.path.isabs(last_filename) else os.path.basename(last_filename))

    # ------------------------------------------------------------------ read
    def load(self, path=None, resume=True, resume_states=True):
        """-> the extra entries of the checkpoint ({} when nothing was loaded or resume_states is False)."""
        if resume and self.has_checkpoint():
            path = self.get_checkpoint_file()  # an existing run overrides the argument
        if not path:
            self._print('No checkpoint found. Initializing model from scratch')
            return {}
        self._print('Loading checkpoint from {}, MD5: {}'.format(path, get_md5(path)))
        checkpoint = self._load_file(path)
        _bare(self.model).load_state_dict(checkpoint.pop('model'))
        if not resume_states:
            return {}
        for key, obj in (('optimizer', self.optimizer), ('scheduler', self.scheduler)):
            if key in checkpoint and obj:
                self._print('Loading {} from {}'.format(key, path))
                obj.load_state_dict(checkpoint.pop(key))
        return checkpoint

    def has_checkpoint(self):
        return os.path.exists(self._tag_file())

    def get_checkpoint_file(self):
        try:
            with open(self._tag_file(), 'r') as f:
                last_saved = f.read()
        except IOError:  # e.g. just deleted by another process
            return ''
        return last_saved if os.path.isabs(last_saved) else os.path.join(self.save_dir, last_saved)

    def _tag_file(self):
        return os.path.join(self.save_dir, self.TAG)

    def _load_file(self, path):
        # checkpoints hold optimizer / scheduler state and plain Python extras, not only tensors
        return torch.load(path, map_location=torch.device('cpu'), weights_only=False)


class CheckpointerV2(Checkpointer):
    """Keeps the last `max_to_keep` checkpoints, like tf.train.Saver (checkpoint.py:122-175)."""

    def __init__(self, *args, max_to_keep=5, **kwargs):
        super(CheckpointerV2, self).__init__(*args, **kwargs)
        self.max_to_keep = max_to_keep
        self._last_checkpoints = []

    def get_checkpoint_file(self):
        try:
            self._last_checkpoints = self._read_tag()
            return self._last_checkpoints[-1]
        except (IOError, IndexError):
            return ''

    def tag_last_checkpoint(self, last_filename):
        self._last_checkpoints = [p for p in self._last_checkpoints if p != last_filename]  # re-used name moves to the end
        self._last_checkpoints.append(last_filename)
        if len(self._last_checkpoints) > self.max_to_keep:
            oldest = self._last_checkpoints.pop(0)
            try:
                os.remove(oldest)
            except Exception as e:
                logging.warning('Ignoring: %s', str(e))
        with open(self._tag_file(), 'w') as f:
            f.write('\n'.join(p if os.path.isabs(p) else os.path.basename(p) for p in self._last_checkpoints))

    def _read_tag(self):
        with open(self._tag_file(), 'r') as f:
            names = [line.rstrip('\n') for line in f.readlines()]
        return [p if os.path.isabs(p) else os.path.join(self.save_dir, p) for p in names if p]
