"""SURVEY.md sec.8f rank 4 on the host: meters, evaluator and checkpoint layout against fixtures written by the imported
reference classes (tests/golden/metrics.npz, tests/golden/checkpoint_ref/; generator: tests/golden/make_golden.py metrics)."""
import json
import os
import shutil

import numpy as np
import pytest
import torch

from mvpnet_amd import checkpoint as CK
from mvpnet_amd import metric as M
from mvpnet_amd.mvpnet3d import SegLoss

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.fixture(scope='module')
def gold():
    return np.load(os.path.join(GOLD, 'metrics.npz'))


def test_meters_and_loss_follow_the_reference(gold):
    acc, iou = M.SegAccuracy(), M.SegIoU(20)
    crit = SegLoss(weight=torch.from_numpy(gold['loss_weight']))
    lines = []
    for it in range(3):
        logit = torch.from_numpy(gold['m%d_logit' % it]).requires_grad_(True)
        label = torch.from_numpy(gold['m%d_label' % it])
        loss = crit({'seg_logit': logit}, {'seg_label': label})['seg_loss']
        loss.backward()
        np.testing.assert_allclose(loss.item(), float(gold['m%d_loss' % it]), rtol=1e-6)
        np.testing.assert_allclose(logit.grad.numpy(), gold['m%d_grad' % it], rtol=1e-5, atol=1e-9)
        acc.update_dict({'seg_logit': logit.detach()}, {'seg_label': label})
        iou.update_dict({'seg_logit': logit.detach()}, {'seg_label': label})
        # the meters' own strings, as the reference's MetricLogger prints them ("name: str(meter)"; the generic logger itself is
        # out of scope -- the reference's accepts these meters through add_meters)
        lines.append('seg_acc: {}  seg_iou: {} || seg_acc: {}  seg_iou: {}'.format(acc, iou, acc.summary_str, iou.summary_str))
        np.testing.assert_allclose([acc.global_avg, acc.avg], gold['m%d_acc' % it], rtol=1e-12)
        np.testing.assert_array_equal(iou.mat.numpy(), gold['m%d_mat' % it])
        np.testing.assert_allclose(iou.iou.numpy(), gold['m%d_iou' % it], rtol=1e-6, equal_nan=True)
    ref_lines = json.loads(str(gold['logger_lines']))           # "seg_acc: ..  seg_iou: ..  loss: ..  lr: .. || <summaries>"
    for mine, ref in zip(lines, ref_lines):
        a, b = ref.split(' || ')
        assert mine == '  '.join(a.split('  ')[:2]) + ' || ' + '  '.join(b.split('  ')[:2])  # character for character
    iou.reset()
    acc.reset()
    assert iou.mat is None and acc.count == 0 and np.isnan(acc.global_avg)


def test_evaluator_follows_the_reference(gold, tmp_path):
    ev, raw = M.Evaluator(M.CLASS_NAMES), M.Evaluator(M.CLASS_NAMES, M.EVAL_CLASS_IDS)
    ids = np.array(M.EVAL_CLASS_IDS + [0])
    for sc in range(3):
        gt, pred = gold['e%d_gt' % sc], gold['e%d_pred' % sc]
        keep = gt.copy()
        ev.update(pred, gt)
        np.testing.assert_array_equal(gt, keep)                 # the caller's labels are left alone
        raw.batch_update([ids[pred]], [np.where(gt >= 0, ids[np.clip(gt, 0, 19)], -100)])
    ev.update(np.zeros(5, np.int64), np.full(5, -100, np.int64))  # nothing valid: skipped
    for name, e in (('ev', ev), ('evraw', raw)):
        np.testing.assert_array_equal(e.confusion_matrix, gold[name + '_cm'])
        np.testing.assert_allclose([e.overall_acc, e.overall_iou], gold[name + '_overall'], rtol=1e-12)
        np.testing.assert_allclose(e.class_iou, gold[name + '_class_iou'], rtol=1e-12, equal_nan=True)
        np.testing.assert_allclose(e.class_seg_acc, gold[name + '_class_acc'], rtol=1e-12, equal_nan=True)
    assert len(M.CLASS_NAMES) == 20 and M.EVAL_CLASS_IDS[-1] == 39


def tiny(seed):
    torch.manual_seed(seed)
    model = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.BatchNorm1d(4), torch.nn.Linear(4, 2))
    opt = torch.optim.Adam(model.parameters(), lr=2e-3)
    return model, opt, torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[2, 4], gamma=0.1)


def test_checkpointer_reads_reference_files_and_keeps_the_layout(tmp_path, capsys):
    work = str(tmp_path / 'run')
    shutil.copytree(os.path.join(GOLD, 'checkpoint_ref'), work)
    expected = np.load(os.path.join(work, 'expected_state.npz'))
    model, opt, sched = tiny(99)
    ck = CK.CheckpointerV2(model, optimizer=opt, scheduler=sched, save_dir=work, max_to_keep=2)
    assert ck.has_checkpoint() and ck.get_checkpoint_file() == os.path.join(work, 'model_000003.pth')
    extra = ck.load('ignored-because-the-run-has-a-tag-file.pth', resume=True)
    assert extra == {'iteration': 3, 'best_metric': pytest.approx(0.3)}
    for k, v in model.state_dict().items():
        np.testing.assert_array_equal(v.numpy(), expected[k])
    assert sched.last_epoch == 3 and opt.state_dict()['state'][0]['step'] == 3           # optimizer / scheduler resumed
    assert opt.param_groups[0]['lr'] == pytest.approx(2e-4)
    # weights only (resume_states=False): nothing else is touched or returned
    m2, o2, s2 = tiny(98)
    assert CK.Checkpointer(m2, o2, s2, save_dir='').load(os.path.join(work, 'model_best.pth'), resume=False, resume_states=False) == {}
    assert s2.last_epoch == 0 and torch.equal(m2[0].weight, model[0].weight)
    # writing: max_to_keep files, tag file with bare names oldest first, extras round-trip
    ck.save('model_000004', iteration=4)
    ck.save('model_000005', iteration=5, note='x')
    assert sorted(os.listdir(work)) == ['expected_state.npz', 'last_checkpoint', 'model_000004.pth', 'model_000005.pth', 'model_best.pth']
    assert open(os.path.join(work, 'last_checkpoint')).read() == os.path.join(work, 'model_000004.pth') + '\n' + os.path.join(work, 'model_000005.pth')
    data = torch.load(os.path.join(work, 'model_000005.pth'), weights_only=False)
    assert sorted(data) == ['iteration', 'model', 'note', 'optimizer', 'scheduler'] and list(data['model']) == list(model.state_dict())
    rest = CK.CheckpointerV2(tiny(1)[0], save_dir=work).load(None)  # no optimizer / scheduler to resume: their states are handed back
    assert sorted(rest) == ['iteration', 'note', 'optimizer', 'scheduler'] and rest['iteration'] == 5 and rest['note'] == 'x'
    # no save_dir: save is a no-op; no checkpoint anywhere: start from scratch
    CK.Checkpointer(model, save_dir='').save('nothing')
    assert CK.CheckpointerV2(model, save_dir=str(tmp_path / 'empty')).load(None) == {}
    assert 'No checkpoint found' in capsys.readouterr().out
    assert len(CK.get_md5(os.path.join(work, 'model_best.pth'))) == 32


def test_checkpointer_relative_dir_and_wrapped_model(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    model, opt, sched = tiny(3)
    wrapped = torch.nn.DataParallel(model)
    ck = CK.Checkpointer(wrapped, optimizer=opt, save_dir='out')
    os.makedirs('out')
    ck.save('model_final', epoch=7)
    assert open('out/last_checkpoint').read() == 'model_final.pth'                        # bare name for a relative directory
    assert list(torch.load('out/model_final.pth', weights_only=False)['model']) == list(model.state_dict())  # no 'module.' prefix
    m2 = tiny(4)[0]
    rest = CK.Checkpointer(m2, save_dir='out').load()
    assert sorted(rest) == ['epoch', 'optimizer'] and rest['epoch'] == 7
    assert torch.equal(m2[2].bias, model[2].bias)
