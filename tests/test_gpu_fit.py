"""The fit loop (ps_vae_amd.train.fit: the Trainer loop of the reference's src/train.py:31-54) end to
end on the HIP model: GPU-collated batches, flat-buffer clip + Adam, LambdaLR per epoch, validation
with the accuracy counters, best-val_loss checkpoint, reload."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('graph_steps', [False, True])
def test_fit_trains_validates_checkpoints_and_reloads(tokenizer, tmp_path, graph_steps):
    from ps_vae_amd import train
    from ps_vae_amd.data.bpe_dataset import get_dataloader
    from ps_vae_amd.data.synthetic import synthetic_step1
    from ps_vae_amd.pl_models import PSVAEModel
    from ps_vae_amd.utils.nn_utils import build_config, default_args
    if not torch.cuda.is_available():
        pytest.fail('GPU tests need a GPU (run with -m "not gpu" on CPU-only machines)')
    dev = torch.device('cuda:0')
    args = default_args()
    torch.manual_seed(5)
    model = PSVAEModel(build_config(args, tokenizer), tokenizer).to(dev)
    mols = synthetic_step1(32 * 9, seed=91, shape='zinc')
    loaders = [get_dataloader(None, tokenizer, 32, shuffle=s, device=dev, data=mols[a:b], reference_layout=False, seed=1)
               for a, b, s in ((0, 32 * 6, True), (32 * 6, 32 * 8, False), (32 * 8, 32 * 9, False))]
    losses = []
    hist = train.fit(model, *loaders, max_epochs=3, grad_clip=10.0, patience=3, save_dir=str(tmp_path),
                     on_step=lambda m, loss: losses.append(float(loss)), graph_steps=graph_steps)
    assert hist['epochs'] == 3 and hist['global_step'] == 18 and len(losses) == 18
    assert hist.get('graph_replays', 0) == (15 if graph_steps else 0)       # all but the three warm-up steps
    assert hist['lr'] == pytest.approx([1e-3, 5e-4, 1e-3 / 3])              # LambdaLR(1 / (epoch + 1))
    assert np.isfinite(losses).all() and np.mean(losses[-6:]) < np.mean(losses[:6])
    assert len(hist['monitor']) == 3 and np.isfinite(hist['monitor']).all() and np.isfinite(hist['test_loss'])
    assert hist['best_path'] is not None and hist['best_score'] == min(hist['monitor'])
    # validation also reports the two accuracies (piece tokens, link classes) as fractions
    model.eval()
    with torch.no_grad():
        model.validation_step(next(iter(loaders[1])), 0)
    for name in ('piece accu', 'edge accu'):
        assert 0.0 <= float(model.logged[name]) <= 1.0
    # the best checkpoint restores the parameters that produced its val_loss
    torch.manual_seed(6)
    fresh = PSVAEModel(build_config(args, tokenizer), tokenizer).to(dev)
    payload = train.load_checkpoint(fresh, hist['best_path'])
    assert payload['val_loss'] == hist['best_score']
    again = train.evaluate(fresh, loaders[1], 'validation_step', 'val_loss')
    # (eps of the reparameterisation is drawn afresh, so the KL-free part agrees only loosely)
    assert abs(again - hist['best_score']) <= 0.05 * abs(hist['best_score'])


def test_main_trains_on_real_qm9_smiles(tmp_path):
    """train.main end to end on real molecules (BASELINE.json configs[0] in small): SMILES files (lines of
    the reference's data/qm9/train.txt, tests/golden/real/) -> process_step1 without RDKit -> packed
    caches beside the files -> two epochs at batch 32 through the recorded step -> checkpoint ->
    PSVAEModel.load_from_checkpoint.  The property scores are zeros (RDKit descriptors)."""
    import os
    from ps_vae_amd import train
    from ps_vae_amd.pl_models import PSVAEModel
    real = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'real')
    with open(os.path.join(real, 'qm9_2048.smi')) as f:
        lines = f.readlines()
    files = {}
    for name, a, b in (('train', 0, 512), ('valid', 512, 640), ('test', 640, 768)):
        d = tmp_path / name                  # one directory per split: the dataset cache lives beside the file
        d.mkdir()
        files[name] = d / f'{name}.txt'
        files[name].write_text(''.join(lines[a:b]))
    np.random.seed(1)
    result = train.main(['--train_set', str(files['train']), '--valid_set', str(files['valid']), '--test_set', str(files['test']),
                         '--vocab', os.path.join(real, 'qm9_vocab_300.txt'), '--save_dir', str(tmp_path / 'ckpt'),
                         '--alpha', '0.1', '--batch_size', '32', '--epochs', '2', '--shuffle'])
    assert result['epochs'] == 2 and result['global_step'] == 32
    assert np.isfinite(result['monitor']).all() and result['monitor'][1] < result['monitor'][0]
    assert np.isfinite(result['test_loss'])
    assert (tmp_path / 'train' / 'processed_data.npz').exists()
    model = PSVAEModel.load_from_checkpoint(result['best_path'])
    assert model.tokenizer.num_piece_type() == 302
