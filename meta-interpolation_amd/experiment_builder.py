"""ExperimentBuilder: train / validate / test driver around SceneAdaptiveInterpolation.

Entry-point surface kept from the reference (experiment_builder.py:9-319): constructed as
``ExperimentBuilder(args, data, model)``; ``run_experiment()`` dispatches on ``--mode``; calls exactly
``model.run_train_iter(data_batch, epoch, do_evaluation)``, ``model.run_validation_iter(data_batch)``,
``model.run_test_iter(data_batch)``, reads ``model.optimizer.param_groups[0]['lr']``, steps
``model.scheduler`` on the validation loss and saves ``{'epoch','arch','state_dict','best_PSNR'}``
checkpoints.  Frames with H*W > 5e5 are evaluated in two halves and stitched (reference :105-115,
:160-169).  Test mode writes each interpolated frame next to its clip and val mode under
``checkpoint/<exp>/<dataset>/...`` like the reference (:194-206, :228-234); tensorboard is out of scope.
"""
import time

import torch

from . import utils


class ExperimentBuilder(object):
    def __init__(self, args, data, model):
        self.args, self.model = args, model
        self.device = torch.device('cuda') if args.cuda else torch.device('cpu')
        self.state = {'best_val_loss': 0., 'best_val_iter': 0, 'current_iter': 0}
        self.total_losses = dict()
        self.best_PSNR = 0
        self.data = data(args=args, current_iter=self.state['current_iter'])
        self.epoch = int(self.state['current_iter'] / args.total_iter_per_epoch)
        self.start_time = time.time()
        self.epochs_done_in_this_run = 0
        if args.resume:
            self.epoch = args.start_epoch
            self.state['current_iter'] = self.epoch * args.total_iter_per_epoch
        self.log = []

    @staticmethod
    def build_loss_summary_string(summary_losses, metrics):
        parts = ["{}: {:.4f}".format(k, float(v)) for k, v in summary_losses.items()
                 if k != 'loss' and 'loss_importance_vector' not in k]
        parts += ["{}: {:.4f}".format(k, float(v.avg)) for k, v in metrics.items()]
        return ", ".join(parts)

    # ---- one iteration of each kind -------------------------------------------------------------
    def train_iteration(self, train_sample, epoch_idx, current_iter, do_evaluation=False):
        images, _ = train_sample
        losses, outputs, metrics = self.model.run_train_iter(data_batch=images, epoch=epoch_idx,
                                                             do_evaluation=do_evaluation)
        return losses, outputs, metrics, current_iter + 1

    def _split(self, frames):
        H, W = frames[0].shape[-2:]
        if H > W:
            return [f[..., :H // 2, :] for f in frames], [f[..., H // 2:, :] for f in frames], -2
        return [f[..., :W // 2] for f in frames], [f[..., W // 2:] for f in frames], -1

    def _eval_frames(self, frames):
        H, W = frames[0].shape[-2:]
        if H * W > 5e5 or (self.args.model == 'rrin' and H * W > 3e5):     # reference :105
            a, b, dim = self._split(frames)
            la, oa = self._eval_frames(a)
            lb, ob = self._eval_frames(b)
            outs = [torch.cat([x, y], dim=dim) for x, y in zip(oa, ob)]
            losses = {k: (la[k] + lb[k]) / 2 for k in la if k in lb}
            return losses, outs
        losses, outputs, _ = self.model.run_validation_iter(data_batch=frames)
        losses['loss'] = losses['loss'].detach()
        return losses, outputs

    def evaluation_iteration(self, val_sample):
        images, _ = val_sample
        losses, outputs = self._eval_frames(images)
        output = outputs[0].squeeze(0).detach()
        target = images[3][0].detach().to(output.device)
        if self.args.model == 'voxelflow':
            target = (target * self.model.std + self.model.mean) / 255.0
        elif self.args.model == 'superslomo':
            target = self.model.revNormalize(target)
        metrics = {'psnr': utils.AverageMeter(), 'ssim': utils.AverageMeter()}
        psnr, ssim = utils.calc_metrics(output, target)
        metrics['psnr'].update(psnr)
        metrics['ssim'].update(float(ssim))
        return losses, outputs, metrics

    def test_iteration(self, test_sample):
        images, _ = test_sample
        H, W = images[0].shape[-2:]
        if H * W > 5e5 or (self.args.model == 'rrin' and H * W > 3e5):
            a, b, dim = self._split(images)
            oa, ob = self.model.run_test_iter(data_batch=a), self.model.run_test_iter(data_batch=b)
            return [torch.cat([x, y], dim=dim) for x, y in zip(oa, ob)]
        return self.model.run_test_iter(data_batch=images)

    # ---- image writers (reference :194-206, :228-234) ---------------------------------------------
    def _write_test_frames(self, imgpaths, outputs):
        """`<data_root>/<prefix>_<mean of the two neighbours' time stamps>.<fmt>` per clip; a right neighbour stamped
        0 counts as 1.0, exactly as the reference does.  Paths that do not carry a time stamp are not written."""
        import os
        for k in range(len(outputs)):
            try:
                f1, f2 = imgpaths[1][k].split('/')[-1], imgpaths[2][k].split('/')[-1]
                t1, t2 = float(f1.split('_')[-1][:-4]), float(f2.split('_')[-1][:-4])
            except (ValueError, IndexError):
                continue
            t2 = 1.0 if t2 == 0 else t2
            utils.save_image(outputs[k], os.path.join(self.args.data_root, '%s_%.06f.%s'
                                                      % (f1.split('_')[0], (t1 + t2) / 2, self.args.img_fmt)))

    def _write_val_frames(self, imgpaths, outputs):
        """`checkpoint/<exp_name>/<dataset>/<a>/<b>/im4.png` for every item of a validation batch."""
        import os
        for k in range(outputs[0].shape[0] if outputs[0].dim() == 4 else len(outputs)):
            paths = imgpaths[3][k].split('/')
            if len(paths) < 3:
                continue
            save_dir = os.path.join('checkpoint', self.args.exp_name, self.args.dataset, paths[-3], paths[-2])
            os.makedirs(save_dir, exist_ok=True)
            out = outputs[0][k] if outputs[0].dim() == 4 else outputs[k].squeeze(0)
            utils.save_image(out, os.path.join(save_dir, paths[-1]))

    # ---- sweeps ------------------------------------------------------------------------------
    def _validation_sweep(self, write_images=False):
        acc = {'psnr': utils.AverageMeter(), 'ssim': utils.AverageMeter()}
        val_losses = {}
        n = self.data.dataset.data_length['val']
        total = int(n / self.args.val_batch_size + 0.99)
        for val_sample in self.data.get_val_batches(total_batches=total):
            losses, outputs, metrics = self.evaluation_iteration(val_sample)
            if write_images:                      # `--mode val` only (reference :228-234)
                self._write_val_frames(val_sample[1]['imgpaths'], outputs)
            for k, v in metrics.items():
                acc[k].update(v.avg, n=v.count)
            for k, v in losses.items():
                if 'loss_importance_vector' in k:
                    continue
                val_losses.setdefault(k, utils.AverageMeter()).update(float(v))
        return {k: v.avg for k, v in val_losses.items()}, acc

    def run_experiment(self):
        args = self.args
        if args.mode == 'test':
            n = self.data.dataset.data_length['test']
            outs = []
            for sample in self.data.get_test_batches(total_batches=int(n / args.test_batch_size)):
                outputs = self.test_iteration(sample)
                outs.append(outputs)
                self._write_test_frames(sample[1]['imgpaths'], outputs)
            print('Test finished: %d clips.' % len(outs))
            return outs
        if args.mode == 'val':
            losses, acc = self._validation_sweep(write_images=True)
            print("%d examples processed" % acc['psnr'].count)
            print("PSNR: %.2f,  SSIM: %.4f\n" % (acc['psnr'].avg, acc['ssim'].avg))
            return losses, acc

        last = int(args.total_iter_per_epoch * args.max_epoch)
        while self.state['current_iter'] < last:
            remaining = last - self.state['current_iter']
            for train_sample in self.data.get_train_batches(total_batches=remaining):
                it = self.state['current_iter']
                losses, _, metrics, self.state['current_iter'] = self.train_iteration(
                    train_sample, epoch_idx=it / args.total_iter_per_epoch, current_iter=it,
                    do_evaluation=(it % args.eval_iter == 0))
                if self.state['current_iter'] % args.log_iter == 1:
                    self.log.append((self.state['current_iter'], float(losses['loss']),
                                     self.model.optimizer.param_groups[0]['lr']))
                    print("iter %d  loss %.6f  %s" % (self.state['current_iter'], float(losses['loss']),
                                                      self.build_loss_summary_string(losses, metrics)), flush=True)
                if self.state['current_iter'] % args.total_iter_per_epoch == 0:
                    val_losses, acc = self._validation_sweep()
                    print("validation PSNR: %.2f,  SSIM: %.4f\n" % (acc['psnr'].avg, acc['ssim'].avg))
                    self.epoch += 1
                    psnr = acc['psnr'].avg
                    is_best = psnr > self.best_PSNR
                    self.best_PSNR = max(psnr, self.best_PSNR)
                    if self.model.task_parallel.rank == 0:
                        utils.save_checkpoint({'epoch': self.epoch, 'arch': args, 'state_dict': self.model.state_dict(),
                                               'best_PSNR': self.best_PSNR}, is_best, args.exp_name)
                    self.model.scheduler.step(val_losses['total'])
                    self.total_losses = dict()
                    self.epochs_done_in_this_run += 1
                if self.state['current_iter'] >= last:
                    break
        return self.log
